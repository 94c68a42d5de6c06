"""CPU: the host-side state machine pieces of the batched pre-optimiser (mft6.py:952-1103)."""
import numpy as np

import common  # noqa: F401
from mcmc_spec_amd import optimizer as opt


def test_bounds_and_step_sizes():
    tlim = [3000.0, 4200.0]
    ok = [np.array([3800.0, 3100.0]), np.array([0.1]), np.array([0.5, 0.4]), np.array([2e-3])]
    assert opt._in_bounds(ok, tlim)
    for group, idx, val in [(0, 0, 4200.0), (0, 1, 3000.0), (1, 0, -1e-3), (2, 0, 1.51), (2, 0, 0.049), (2, 1, 1.0),
                            (2, 1, 0.05), (3, 0, 0.1), (3, 0, 1 / 3000)]:
        bad = [v.copy() for v in ok]
        bad[group][idx] = val
        assert not opt._in_bounds(bad, tlim), (group, idx, val)
    coarse = opt._step_sizes(2, [0.6, 0.45], 2.2e-3, False)
    fine = opt._step_sizes(2, [0.6, 0.45], 2.2e-3, True)
    assert coarse[0] == [250.0, 250.0] and coarse[1] == [0.05] and np.allclose(coarse[2], [0.06, 0.045])
    assert np.isclose(coarse[3][0], 0.02 * 2.2e-3) and fine[0] == [20.0, 20.0] and np.isclose(fine[3][0], 0.005 * 2.2e-3)
    assert opt._step_sizes(3, [0.6, 0.4, 0.3], 2e-3, False)[3][0] == 0.05 * 2e-3


def test_repair_loop_counts_like_the_reference():
    tlim = [3000.0, 4200.0]
    # Teff 250 K under the floor needs 3 x +100; secondary above primary afterwards needs 1 x -100
    var = [np.array([2750.0, 3060.0]), np.array([0.1]), np.array([0.5, 0.4]), np.array([2e-3])]
    assert opt._repair_count(var, tlim, 10, 10_000) == 10 + 1 + 3 + 1
    # A_V = -0.25 -> three +0.1 steps; radius 0.02 -> three +0.01 steps
    var = [np.array([3800.0, 3100.0]), np.array([-0.25]), np.array([0.02, 0.4]), np.array([2e-3])]
    assert opt._repair_count(var, tlim, 0, 10_000) == 1 + 3 + 3
    # the budget cap stops the loops
    var = [np.array([1000.0, 900.0]), np.array([0.1]), np.array([0.5, 0.4]), np.array([2e-3])]
    assert opt._repair_count(var, tlim, 95, 100) == 100


def test_opt_prior_forms():
    assert opt.opt_prior_one(np.array([0.3]), 0.1, 0.05) == ((0.3 - 0.1) / 0.05) ** 2
    assert opt.opt_prior_sum([0.5, 0.4], [0.45, 0.0], [0.05, 0.1]) == ((0.5 - 0.45) / 0.05) ** 2  # p == 0 skipped
