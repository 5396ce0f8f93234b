"""bench.py's contract (the driver's entry point): the plain `--gpus N` call without a launcher runs the in-library device
group (RCCL inside libmmcmc.so) instead of refusing, the JSON line carries the contract's keys, and `roofline.frac` is the
HBM fraction recomputable from its own fields."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                         timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]  # RCCL prints a version banner on stdout
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_decides_group_before_touching_a_gpu():
    # no GPU here: both spellings must reach the "needs an MI355X" exit of main_group, not the launcher refusal
    for args in (["--gpus", "2", "--group"], ["--gpus", "2"]):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env)
        import torch

        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
        assert out.returncode != 0 and "needs an MI355X" in (out.stdout + out.stderr), out.stderr[-500:]
        assert "torch.distributed.run" not in out.stderr


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_a_scaling_line_that_fell_back_cannot_exit_zero():
    """over N DISTINCT devices the diagnostics must have travelled over RCCL with N ranks and the ordinals
    must be N different GPUs; otherwise the line says so in capitals and the process exits non-zero AFTER printing it.
    Fed here with the statuses a box without RCCL (-1), with a failing ncclCommInitAll (-2), a short communicator and a
    GPU listed under two ordinals would produce -- no GPU needed."""
    B = _bench_module()
    ids8 = [f"0000:{b:02x}:00.0" for b in range(8)]
    data, code, problems = B.group_verdict(list(range(8)), ids8, 1, 8)
    assert (data, code, problems) == ("synthetic", 0, [])
    for status in (-1, -2, 0):
        data, code, problems = B.group_verdict(list(range(8)), ids8, status, 0)
        assert code != 0 and "DIAGNOSTICS FELL BACK TO THE HOST" in data and "NOT A VALID SCALING MEASUREMENT" in data
        assert len(problems) == 2  # the fallback and the missing ranks
    data, code, _ = B.group_verdict(list(range(8)), ids8, 1, 4)
    assert code != 0 and "RCCL COMMUNICATOR HAS 4 RANKS FOR 8 DEVICES" in data
    data, code, _ = B.group_verdict([0, 1], ["0000:05:00.0", "0000:05:00.0"], 1, 2)
    assert code != 0 and "SHARE A PCI BUS ID" in data
    # the one-GPU rehearsal (a device listed twice) goes through the host BY DESIGN: exit 0, and the line says what it is
    data, code, problems = B.group_verdict([0, 0], ["0000:05:00.0"] * 2, 0, 0)
    assert code == 0 and problems == [] and "REHEARSAL" in data
    assert B.group_verdict([0], ["0000:05:00.0"], 1, 1) == ("synthetic", 0, [])


def test_a_group_line_with_a_problem_is_printed_and_then_exits_non_zero(monkeypatch, capsys):
    """main_group end to end with the device group replaced by a stand-in whose exchange status is the -1 / -2 fallback:
    the JSON line IS printed (it is the evidence) and SystemExit carries a non-zero code."""
    import types

    import numpy as np

    B = _bench_module()
    for status in (-1, -2):
        class FakeGroup:
            def __init__(self, target, init, eps, L, devices):
                self.devices, self.exchange_status = list(devices), status

            def set_seed(self, seed):
                return self

            def exchange(self):
                return status, 0

            def run_async(self, nc, nd):
                pass

            def sync(self):
                pass

            def timer_start(self):
                pass

            def timer_stop(self):
                return np.array([0.19, 0.20], dtype=np.float32) * 2

            def split_rhat_mean_ess(self):
                return np.full(3, 0.6, np.float32), np.full(3, 1e4, np.float32)

            def stats_phases(self):
                return {"local_partials_ms": 0.1, "exchange_ms": 2.0, "finish_ms": 0.01}

            def pci_bus_ids(self):
                return ["0000:05:00.0", "0000:15:00.0"]

            def close(self):
                pass

        import mini_mcmc_amd.group as G
        import torch

        monkeypatch.setattr(G, "HMCGroup", FakeGroup)
        monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
        monkeypatch.delenv("MMCMC_BENCH_GROUP_DEVICES", raising=False)
        args = types.SimpleNamespace(gpus=2, steps=2, warmup=1, preroll_seconds=0.0)
        with pytest.raises(SystemExit) as e:
            B.main_group(args)
        assert e.value.code not in (0, None)
        line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
        assert len(line) == 1
        j = json.loads(line[0])
        assert "DIAGNOSTICS FELL BACK TO THE HOST" in j["data"] and j["rccl_ranks"] == 0 and len(j["problems"]) == 2
        assert j["pci_bus_ids"] == ["0000:05:00.0", "0000:15:00.0"] and j["kernel_ms_spread_max_over_min"] == pytest.approx(0.20 / 0.19, rel=1e-5)
        assert j["stats_phases_ms"]["exchange_ms"] == pytest.approx(2.0)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--gpus", "2", "--group"], ["--gpus", "2"]])
def test_bench_group_path_on_one_device(args):
    """Two shards on device 0 through the host exchange (RCCL refuses one device twice): keeps the plain multi-GPU call alive."""
    j = _run(args + ["--steps", "3", "--warmup", "1", "--preroll-seconds", "0"], {"MMCMC_BENCH_GROUP_DEVICES": "0,0"})
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["unit"] == "samples/s"
    assert j["value"] > 1e9 and j["config"]["devices"] == [0, 0]
    assert j["diagnostics_exchange"].startswith("host (a device is listed twice)") and j["rccl_ranks"] == 0
    assert "REHEARSAL" in j["data"] and j["problems"] == [] and len(j["pci_bus_ids"]) == 2 and j["pci_bus_ids"][0] == j["pci_bus_ids"][1]
    assert j["kernel_ms_spread_max_over_min"] >= 1.0 and set(j["stats_phases_ms"]) == {"local_partials_ms", "exchange_ms", "finish_ms"}
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1


@pytest.mark.gpu
def test_group_line_measures_the_kernel_like_the_plain_line():
    """The in-library device group must not put a host round trip between steps: on one device `--gpus 1 --group` is within
    3 % of the plain N = 1 line, and two shards on the one device within a few per cent of twice it.
    Best of three each: the two calls are separate processes on a box whose clocks move by a per cent or two."""
    common = ["--steps", "300", "--warmup", "10", "--no-side", "--no-cpu-baseline"]
    plain = min(_run(common)["ms_per_step"] for _ in range(3))
    g1 = [_run(common + ["--gpus", "1", "--group"]) for _ in range(3)]
    group = min(j["ms_per_step"] for j in g1)
    assert group <= plain * 1.03, (group, plain)
    j = g1[0]
    assert j["rccl_ranks"] == 1 and j["diagnostics_exchange"].startswith("rccl") and len(j["kernel_ms_per_device"]) == 1
    assert j["roofline"]["kernel_ms"] <= j["ms_per_step"] * 1.02
    two = min(_run(common + ["--gpus", "2", "--group"], {"MMCMC_BENCH_GROUP_DEVICES": "0,0"})["ms_per_step"] for _ in range(3))
    assert two <= 2 * plain * 1.05, (two, plain)


@pytest.mark.gpu
def test_bench_line_contract_and_roofline_is_the_hbm_fraction():
    j = _run(["--steps", "5", "--warmup", "2", "--no-side", "--cpu-seconds", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["dtype"] == "f32" and j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0
    alg = 65536 * 3 * 4 * (400 + 2)
    assert r["algorithmic_bytes_per_launch"] == alg
    assert abs(r["achieved"] - alg / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["kernel_ms"] <= j["ms_per_step"] * 1.02
    # counter-derived fields are either quoted with their provenance or absent -- never literals
    if r["issue"] is not None:
        c = r["issue"]["counters"]
        assert c["round"].startswith("r6") and "mm_run_split_kernel" in c["kernel"] and len(c["sources_sha256"]) == 64
    if r["traffic"] is not None:
        assert 0.9 < r["traffic"] / alg < 1.5 and "profiles/hmc_kernel_traffic.json" in r["traffic_source"]
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["single_thread_samples_per_s"] > 1e5
    # the all-threads rate per thread must be explained if it is under half the single-thread rate
    assert cb["per_thread_over_single_thread"] >= 0.5 or "per_thread_note" in cb
    assert ("ess_per_s" in j) != ("ess_per_s_unconverged" in j)
    assert ("ess_per_s" in j) == (j["split_rhat_max_conventional"] <= 1.05)


@pytest.mark.gpu
def test_bench_under_the_launcher_two_ranks_rehearsed_on_one_device():
    """The driver's N > 1 call -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, one rank per
    GPU -- rehearsed with both ranks on device 0 over gloo (MMCMC_BENCH_DRYRUN_ONE_DEVICE=1: RCCL refuses one device twice);
    the line says it is a dry run.  Keeps the rank bookkeeping (shards, barriers, max over ranks, rank 0 prints) alive."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MMCMC_BENCH_DRYRUN_ONE_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--preroll-seconds", "0", "--no-side"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and "DRY RUN" in j["data"]
    assert j["config"]["parallelism"].startswith("chains sharded x2")
    assert "cpu_baseline" not in j  # the contract times the CPU baseline on rank 0 at N = 1 only
