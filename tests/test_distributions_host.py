"""Host-side pieces of distributions.rs that are not on the GPU path: the reference's own tests restated
(distributions.rs:621-832)."""
import math

import numpy as np


def test_gaussian2d_normalized_logp():
    from mini_mcmc_amd.distributions import Gaussian2D

    # test_gaussian2d_logp (distributions.rs:812-831)
    g = Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]])
    assert abs(g.logp([0.5, -0.5]) - (-2.0878770664093453)) < 1e-10
    g = Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
    x = np.array([0.3, -0.2])
    d = x - np.array([0.0, 1.0])
    S = np.array([[4.0, 2.0], [2.0, 3.0]])
    ref = -math.log(2 * math.pi) - 0.5 * math.log(np.linalg.det(S)) - 0.5 * d @ np.linalg.inv(S) @ d
    assert abs(g.logp(x) - ref) < 1e-12


def test_categorical_reference_cases():
    from mini_mcmc_amd.distributions import Categorical

    cat = Categorical([0.2, 0.3, 0.5], seed=1)
    # test_categorical_logp_f64 (:621-665) and the Target impl (:779-810)
    for i, p in enumerate((0.2, 0.3, 0.5)):
        assert abs(cat.logp(i) - math.log(p)) < 1e-7
    assert cat.logp(3) == float("-inf") and cat.unnorm_logp([1]) == cat.logp(1) and cat.unnorm_logp([3]) == float("-inf")
    # test_categorical_sampling_f64 (:668-697): 100 000 draws within 1 % absolute
    counts = np.bincount([cat.sample() for _ in range(100_000)], minlength=3) / 100_000
    assert np.all(np.abs(counts - [0.2, 0.3, 0.5]) < 0.01)
    # normalisation on construction, single value (:764-777)
    assert np.allclose(Categorical([2.0, 3.0, 5.0]).probs, [0.2, 0.3, 0.5])
    assert Categorical([1.0]).sample() == 0
