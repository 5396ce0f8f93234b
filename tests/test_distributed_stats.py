"""The multi-GPU path on the CPU: two processes, gloo backend, 127.0.0.1.  Each rank holds a shard of the chains'
sufficient statistics (computed here from a host sample the way the GPU kernel defines them) and the exchange in
mini_mcmc_amd.stats.gather_partials (all-gather + all-reduce) followed by the host finish must reproduce the
single-process answer of the oracle.  The data-path of sampling itself has no collective (chains are independent;
tests/test_gpu_parity.py::test_results_independent_of_launch_partition_and_sharding covers the chain-offset keying)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _host_partials(x):
    """means [2, C, D], ssq [2, C, D], acov_sum [m, D] of a local sample x [C, n, D] (definition of the GPU kernel)."""
    c, n, d = x.shape
    m = n // 2
    halves = np.stack([x[:, :m], x[:, n - m:]], axis=0).astype(np.float64)  # [2, C, m, D]
    mu = halves.mean(axis=2)
    y = halves - mu[:, :, None, :]
    ssq = (y**2).sum(axis=2)
    ac = np.stack([(y[:, :, : m - lag] * y[:, :, lag:]).sum(axis=2).sum(axis=(0, 1)) for lag in range(m)])
    return mu.astype(np.float32), ssq.astype(np.float32), ac.astype(np.float32)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from mini_mcmc_amd import stats as S

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    c_total, n, d = 12, 120, 3
    x = np.zeros((c_total, n, d), dtype=np.float32)
    e = rng.standard_normal((c_total, n, d)).astype(np.float32)
    for t in range(1, n):
        x[:, t] = 0.5 * x[:, t - 1] + e[:, t]
    x[:, :, 1] += np.arange(c_total)[:, None] * 0.2
    cl = c_total // world
    mu, ssq, ac = _host_partials(x[rank * cl:(rank + 1) * cl])
    g_mu, g_ssq, g_ac = S.gather_partials(torch.from_numpy(mu), torch.from_numpy(ssq), torch.from_numpy(ac))
    rhat, ess = S.stats_finish(g_mu, g_ssq, g_ac)
    # the exchange without a gather (two small all-reduces) must agree with the gathered one
    dsum, wsum, r_ac, c2 = S.reduce_partials(torch.from_numpy(mu), torch.from_numpy(ssq), torch.from_numpy(ac))
    rhat2, ess2 = S.stats_finish_sums(dsum, wsum, r_ac, c2)
    assert c2 == 2 * c_total
    np.testing.assert_allclose(rhat2, rhat, rtol=2e-6)
    np.testing.assert_allclose(ess2, ess, rtol=2e-5)
    q.put((rank, rhat, ess, g_mu.shape))
    dist.destroy_process_group()


def test_two_rank_gather_and_finish_matches_single_process(O):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    # regenerate the same global sample and compare with the oracle's single-process result
    rng = np.random.default_rng(123)
    c_total, n, d = 12, 120, 3
    x = np.zeros((c_total, n, d), dtype=np.float32)
    e = rng.standard_normal((c_total, n, d)).astype(np.float32)
    for t in range(1, n):
        x[:, t] = 0.5 * x[:, t - 1] + e[:, t]
    x[:, :, 1] += np.arange(c_total)[:, None] * 0.2
    r0, e0 = O.split_rhat_mean_ess(x)
    for rank, rhat, ess, shape in res:
        assert shape == (2 * c_total, d)
        np.testing.assert_allclose(rhat, r0, rtol=5e-5)
        np.testing.assert_allclose(ess, e0, rtol=2e-3)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])  # every rank, same answer
