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
    return json.loads(out.stdout.strip().splitlines()[-1])


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


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--gpus", "2", "--group"], ["--gpus", "2"]])
def test_bench_group_path_on_one_device(args):
    """Two shards on device 0 through the host exchange (RCCL refuses one device twice): keeps the plain multi-GPU call alive."""
    j = _run(args + ["--steps", "3", "--warmup", "1", "--preroll-seconds", "0"], {"MMCMC_BENCH_GROUP_DEVICES": "0,0"})
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["unit"] == "samples/s"
    assert j["value"] > 1e9 and j["config"]["devices"] == [0, 0]
    assert j["diagnostics_exchange"].startswith("host (a device is listed twice)") and j["rccl_ranks"] == 0
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1


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
        assert c["round"].startswith("r4") and "mm_run_split_kernel" in c["kernel"] and len(c["sources_sha256"]) == 64
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
