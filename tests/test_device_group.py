"""Device groups behind the C ABI (csrc/mm_group.hip, include/mmcmc.h: mmcmc_hmc_group_*, mmcmc_mh_group_*): one call
runs every chain on N GPUs (ChainRunner::run core.rs:176-186: "run executes all chains"), split-R-hat / ESS reduced
inside the library over RCCL.

The GPU box has ONE device, so what runs there is (a) a one-device group through RCCL (ncclCommInitAll /
ncclAllGather / ncclAllReduce from libmmcmc.so itself) and (b) several shards on device 0 (devices = [0, 0, 0]: every
sharding path -- chain offsets, output slices, unequal shards, assembling the statistics -- with the exchange through
the host, since RCCL refuses a device twice).  Both must reproduce the single-handle run bit for bit, and the
diagnostics the single-GPU entry point's to float rounding of the lag sums.  A real multi-device test is included
and skips below two devices; no scaling curve has been measured yet."""
import ctypes as C

import numpy as np
import pytest


def test_group_symbols_exported_and_no_device_without_gpu():
    import torch

    import mini_mcmc_amd
    from mini_mcmc_amd import _lib as L

    lib = mini_mcmc_amd.lib()
    for sym in ("create", "seed", "set_chain_offset", "run", "state", "split_rhat_mean_ess", "sync", "stream_timer", "exchange",
                "stats_phases", "destroy"):
        assert hasattr(lib, "mmcmc_hmc_group_" + sym) and hasattr(lib, "mmcmc_mh_group_" + sym) and hasattr(lib, "mmcmc_nuts_group_" + sym)
    assert hasattr(lib, "mmcmc_hmc_group_run_async") and hasattr(lib, "mmcmc_mh_group_run_async") and hasattr(lib, "mmcmc_device_pci_bus_id")
    assert not hasattr(lib, "mmcmc_nuts_group_run_async")  # NUTS hands adaptation state over between launches: blocking only
    assert lib.mmcmc_version() >= 101  # the version that made "asynchronous" an entry point of its own
    assert b"broken" in lib.mmcmc_status_string(L.ERR_GROUP_BROKEN)
    if not torch.cuda.is_available():
        from mini_mcmc_amd.distributions import RosenbrockND

        h = C.c_void_p()
        d = RosenbrockND(3).desc()
        init = np.zeros((4, 3), dtype=np.float32)
        dev = (C.c_int * 1)(0)
        st = lib.mmcmc_hmc_group_create(C.byref(h), C.byref(d), init.ctypes.data, 4, 0.03, 10, L.F32, dev, 1)
        assert st == L.ERR_NO_DEVICE  # no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_group_reproduces_single_handle_run(O, devices):
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND
    from mini_mcmc_amd.group import HMCGroup, MetropolisHastingsGroup
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    C_, nc, nd = 1000, 120, 30  # 1000 = 3 * 333 + 1: unequal shards
    init = init_with_seed(C_, 3, 42, np.float32)
    one = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42)
    ref = one.run(nc, nd)
    g = HMCGroup(RosenbrockND(3), init, 0.032, 10, devices=devices).set_seed(42)
    out = g.run(nc, nd)
    assert np.array_equal(out, ref) and np.array_equal(g.accept_counts, one.accept_counts)
    assert np.array_equal(g.state(), one.state())
    sh = g.shards()
    assert [s[1] for s in sh] == [sum(x[2] for x in sh[:i]) for i in range(len(sh))] and sum(s[2] for s in sh) == C_
    assert all(s[3] for s in sh)  # device-resident shards of the sample
    r0, e0 = S.split_rhat_mean_ess(ref)
    r1, e1 = g.split_rhat_mean_ess()
    assert g.used_rccl == (len(devices) == 1)  # RCCL (ncclAllGather + ncclAllReduce inside libmmcmc.so) / host exchange
    np.testing.assert_allclose(r1, r0, rtol=2e-6)
    np.testing.assert_allclose(e1, e0, rtol=1e-4)
    ro, eo = O.split_rhat_mean_ess(ref)
    np.testing.assert_allclose(r1, ro, rtol=1e-4)
    np.testing.assert_allclose(e1, eo, rtol=5e-3)
    # a second run continues the chains; to_host=False keeps the sample on the devices
    ref2 = one.run(10, 0)
    g.run(10, 0, to_host=False)
    assert np.array_equal(g.state(), ref2[:, -1, :])
    # MH group, f64, with a chain offset
    init2 = init_with_seed(257, 2, 7, np.float64)
    tgt = Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
    m1 = MetropolisHastings(tgt, IsotropicGaussian(1.0), init2).seed(5).set_chain_offset(1 << 33)
    refm = m1.run(64, 8)
    mg = MetropolisHastingsGroup(tgt, IsotropicGaussian(1.0), init2, devices=devices).seed(5).set_chain_offset(1 << 33)
    assert np.array_equal(mg.run(64, 8), refm) and np.array_equal(mg.accept_counts, m1.accept_counts)
    rm, em = mg.split_rhat_mean_ess()
    rs, es = S.split_rhat_mean_ess(refm)
    np.testing.assert_allclose(rm, rs, rtol=2e-6)
    np.testing.assert_allclose(em, es, rtol=1e-4)


@pytest.mark.gpu
def test_group_over_two_real_devices_uses_rccl():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the test box has one); covered structurally by the [0, 0] group")
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.group import HMCGroup
    from mini_mcmc_amd.hmc import HMC

    n_dev = torch.cuda.device_count()
    init = init_with_seed(4096 * n_dev, 3, 42, np.float32)
    ref = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42).run(100, 20)
    g = HMCGroup(RosenbrockND(3), init, 0.032, 10, devices=list(range(n_dev))).set_seed(42)
    assert np.array_equal(g.run(100, 20), ref)
    r1, e1 = g.split_rhat_mean_ess()
    assert g.used_rccl
    r0, e0 = S.split_rhat_mean_ess(ref)
    np.testing.assert_allclose(r1, r0, rtol=2e-6)
    np.testing.assert_allclose(e1, e0, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_nuts_group_reproduces_single_handle_run(O, devices):
    """mmcmc_nuts_group_*: NUTS::run of every chain from one call; shards keyed by the global chain index, so the sample,
    tree shapes and positions equal the single-handle run bit for bit (one-chain-per-lane kernel on RosenbrockND(3), the
    lane-group / scheduler kernel on the 32-D Gaussian of config 5), and the diagnostics match the single-GPU entry."""
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import GaussianND, RosenbrockND
    from mini_mcmc_amd.group import NUTSGroup
    from mini_mcmc_amd.nuts import NUTS

    for tgt, mode, C_, scale in ((RosenbrockND(3), 0, 1000, 0.5), (GaussianND.ill_conditioned(32, 100.0, 5), 2, 700, 0.1)):
        init = init_with_seed(C_, tgt.dim, 42) * scale
        for progress in (False, True):
            one = NUTS(tgt, init, 0.8, mode=mode).set_seed(42)
            ref = one._run(40, 24, progress, "numpy")
            g = NUTSGroup(tgt, init, 0.8, mode=mode, devices=devices).set_seed(42)
            out = g.run(40, 24, progress=progress)
            assert np.array_equal(out, ref), (type(tgt).__name__, progress)
            assert np.array_equal(g.state(), one.positions()) and np.array_equal(g.leapfrog_counts(), one.leapfrog_counts())
            r0, e0 = S.split_rhat_mean_ess(ref)
            r1, e1 = g.split_rhat_mean_ess()
            assert g.used_rccl == (len(devices) == 1)
            np.testing.assert_allclose(r1, r0, rtol=2e-4)
            np.testing.assert_allclose(e1, e0, rtol=2e-3)
            g.close()


@pytest.mark.gpu
def test_config4_shape_on_one_device(O):
    """BASELINE config 4 at full size -- 3-D Rosenbrock HMC, 8 x 65 536 = 524 288 chains, run(400, 50) -- with its eight
    shards on the ONE device of the test box (no 8-GPU node here): the sharding, the global-index keying and the
    reduction over all 1 048 576 half-chains at the real sizes.  A shard must equal a single handle given the shard's
    slice of the initial positions and its chain offset, the first chains of a shard the host build, and the diagnostics
    of the group the single-GPU entry point's on the gathered sample (rounding of the lag sums apart)."""
    import torch

    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.group import HMCGroup
    from mini_mcmc_amd.hmc import HMC

    n_shards, per = 8, 65536
    C_ = n_shards * per
    init = init_with_seed(C_, 3, 42, np.float32)
    g = HMCGroup(RosenbrockND(3), init, 0.032, 10, devices=[0] * n_shards).set_seed(42)
    g.run(400, 50, to_host=False)
    sh = g.shards()
    assert [s[2] for s in sh] == [per] * n_shards and [s[1] for s in sh] == [per * i for i in range(n_shards)]
    # shard 5 == a single handle on that slice with that offset (bit for bit, all 65 536 chains)
    k = 5
    one = HMC(RosenbrockND(3), init[k * per:(k + 1) * per], 0.032, 10).set_seed(42).set_chain_offset(k * per)
    ref = one.run(400, 50, to="torch")
    import ctypes as C

    shard = torch.empty_like(ref)
    assert torch.cuda.current_device() == sh[k][0]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(shard.data_ptr(), sh[k][3], ref.numel() * 4, 4) == 0  # hipMemcpyDefault, device to device
    torch.cuda.synchronize()
    assert torch.equal(shard, ref)
    # its first chains == the host build of the engine
    twin, _, _ = O.engine_host_run("hmc", O.ROSENBROCK_ND, 3, [], init[k * per:k * per + 64], 0.032, 400, 50, seed=42, n_leapfrog=10,
                                   dtype=np.float32, chain_offset=k * per)
    assert np.array_equal(ref[:64].cpu().numpy(), twin)
    # diagnostics over all 524 288 chains (host exchange: a device listed eight times)
    rhat, ess = g.split_rhat_mean_ess()
    assert not g.used_rccl and np.all(np.isfinite(rhat)) and np.all(ess > 1e5)
    st = g.state()
    assert st.shape == (C_, 3) and np.array_equal(st[k * per:(k + 1) * per], one.state())


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_group_async_runs_equal_the_blocking_ones(devices):
    """A run that hands nothing back to the host is only enqueued on the shards' streams (include/mmcmc.h): several of them
    back to back, then the diagnostics, state and a blocking run, must give what the blocking calls give -- the streams
    order everything -- and the exchange path is known from creation on (the RCCL communicators are made there)."""
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.group import HMCGroup
    from mini_mcmc_amd.hmc import HMC

    C_, nc, nd = 4099, 200, 20
    init = init_with_seed(C_, 3, 42, np.float32)
    a = HMCGroup(RosenbrockND(3), init, 0.032, 10, devices=devices).set_seed(42)
    b = HMCGroup(RosenbrockND(3), init, 0.032, 10, devices=devices).set_seed(42)
    status, ranks = a.exchange()
    assert (status, ranks) == ((1, 1) if len(devices) == 1 else (0, 0))
    a.timer_start()
    for k in range(5):
        assert a.run_async(nc, nd if k == 0 else 0) is None  # enqueued only
    ms = a.timer_stop()
    assert ms.shape == (len(devices),) and (ms > 0).all()
    for k in range(5):
        ref = b.run(nc, nd if k == 0 else 0)  # blocking, host copy
    ra, ea = a.split_rhat_mean_ess()  # ordered behind the five queued runs by the shards' streams
    rb, eb = b.split_rhat_mean_ess()
    assert np.array_equal(ra, rb) and np.array_equal(ea, eb) and a.exchange_status == status
    assert np.array_equal(a.state(), b.state()) and np.array_equal(a.state(), ref[:, -1, :])
    # a growing sample buffer while launches are still queued on the old one, then a blocking run behind the queue
    a.run_async(nc, 0)
    a.run_async(2 * nc, 0)
    b.run(nc, 0, to_host=False)
    b.run(2 * nc, 0, to_host=False)
    assert np.array_equal(a.run(7, 0), b.run(7, 0)) and np.array_equal(a.accept_counts, b.accept_counts)
    a.sync()
    one = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42)
    one.run(5 * nc + 3 * nc + 7, nd)
    assert np.array_equal(a.state(), one.state())


@pytest.mark.gpu
def test_group_run_with_nothing_handed_back_blocks_and_async_is_its_own_entry_point():
    """version 100 inferred "asynchronous" from out_host == NULL && accept_counts == NULL, so a C caller who
    passed NULL / NULL and then read the shard's device pointer on its own stream raced the queued kernels.  Now
    mmcmc_*_group_run blocks whatever its arguments -- checked by reading the shard's device memory on ANOTHER stream with no
    synchronisation of ours right after the call returns -- and the queued spelling is mmcmc_*_group_run_async."""
    import torch

    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.group import HMCGroup
    from mini_mcmc_amd.hmc import HMC

    C_, nc, nd = 65536, 400, 50  # the headline launch: long enough (0.19 ms) that an unsynchronised read would see old bytes
    init = init_with_seed(C_, 3, 42, np.float32)
    g = HMCGroup(RosenbrockND(3), init, 0.032, 10, devices=[0]).set_seed(42)
    one = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42)
    ref = one.run(nc, nd)
    side = torch.cuda.Stream()
    for rep in range(3):
        assert g.run(nc, nd if rep == 0 else 0, to_host=False, accept_counts=False) is None  # NULL / NULL: must BLOCK
        (dev, first, n, ptr), = g.shards()
        host = np.empty((C_, nc, 3), dtype=np.float32)
        from mini_mcmc_amd import _lib as L

        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        assert hip.hipMemcpyAsync(host.ctypes.data, ptr, host.nbytes, 2, C.c_void_p(side.cuda_stream)) == 0
        side.synchronize()
        if rep == 0:
            assert np.array_equal(host, ref)
        else:
            assert np.array_equal(host, one.run(nc, 0))
    assert g.pci_bus_ids()[0].count(":") == 2
    r, e = g.split_rhat_mean_ess()
    ph = g.stats_phases()
    assert set(ph) == {"local_partials_ms", "exchange_ms", "finish_ms"} and all(v > 0 for v in ph.values()) and sum(ph.values()) < 50


@pytest.mark.gpu
def test_group_cross_chain_sums_keep_their_digits_on_an_offset_target(O):
    """the RCCL finish forms B from sum(mean^2) - sum(mean)^2 / 2C over f32 means; unshifted, a parameter far from
    the origin gives digits away to cancellation (f64 partial sums: ~1e-7 relative at this offset, more further out, where
    an f32 sampler no longer resolves its target anyway).  The device reduction now subtracts the shift the single-device
    tail kernel uses (the mean of the first half of global chain 0) before squaring.  Gaussian2D centred at (3000, -2000)
    with unit covariance: the group over RCCL (one rank: the path with the device reduction) must give the single-device
    R-hat / ESS and the oracle's."""
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian
    from mini_mcmc_amd.group import MetropolisHastingsGroup

    C_, nc, nd = 8192, 400, 100
    mean = np.array([3000.0, -2000.0])
    init = (init_with_seed(C_, 2, 42, np.float32).astype(np.float64) + mean).astype(np.float32)
    g = MetropolisHastingsGroup(Gaussian2D(mean.tolist(), [[1.0, 0.0], [0.0, 1.0]]), IsotropicGaussian(1.0), init, devices=[0]).seed(42)
    out = g.run(nc, nd)
    r1, e1 = g.split_rhat_mean_ess()
    assert g.exchange_status == 1  # RCCL with one rank: mm_group_cross_sums_kernel
    r0, e0 = S.split_rhat_mean_ess(out)
    np.testing.assert_allclose(r1, r0, rtol=1e-5)
    np.testing.assert_allclose(e1, e0, rtol=1e-4)
    ro, eo = O.split_rhat_mean_ess(out)
    np.testing.assert_allclose(r1, ro, rtol=1e-4)
    np.testing.assert_allclose(e1, eo, rtol=5e-3)
    assert np.all(np.abs(r1 - 1.0) < 0.05)  # a converged unit Gaussian: with the digits lost this is far off or NaN
