"""Stated parity tolerances (fp64).

PINV_RTOL bounds  max_j |dq_hip - dq_oracle| / (1 + max_j |dq_oracle|)  per
instance.  Rationale: the reference stacks the first EqualityConstraint twice
(pseudo_inverse.py:317-326 + :382-396), so every lower-priority projector
solves with (2 J'J + lam I) whose condition number is ~2 sigma_max^2 / lam
~ 1e8..1e9 at the default lam = 1e-7 (:55-56).  Two correct fp64 evaluations of
that projector (LU vs Gaussian elimination vs LDL^T, different summation
order) differ by up to ~1e-9 relative - measured between the two independent
CPU oracles in tests/test_oracle.py - and CasADi's own linear solver is a
third such evaluation.  1e-7 leaves two orders of magnitude of margin over
that noise while still rejecting the "textbook" algorithm (no double
processing), which is off by >= 1e-4 (tests/test_oracle.py::test_quirk_matters).
Well-conditioned paths (single task, wide solves) agree to ~1e-12 and are
asserted at PINV_RTOL_TIGHT.

QP_RTOL: the QP optimum is unique (H diagonal > 0); the device active-set
solve and the oracle agree to ~1e-10; KKT residuals are asserted separately.
"""
PINV_RTOL = 1e-7
# conditioning-aware bound: where the smallest singular value of the chain's geometric
# Jacobian is >= 1e-2 the damped solve of a task with m <= n rows is well conditioned relative to
# lam = 1e-7 and two correct fp64 evaluations agree to ~1e-11; those instances are held to 1e-9, the
# others to PINV_RTOL.  This applies to skills WITHOUT a lower-priority equality behind the first one:
# behind it the reference projects through the doubly stacked Jacobian [J; J] (tall branch, 2J'J + lam I,
# n x n of rank <= m < n), whose condition number is ~2 sigma_max^2 / lam ~ 1e8 for EVERY configuration,
# so stacks are held to PINV_RTOL throughout (measured against the reference's own code run on the
# stand-in casadi: <= 1.6e-9, tests/test_refpins.py).
PINV_RTOL_WELL = 1e-9
SIGMA_WELL = 1e-2


def pinv_rtol(sigma_min, stacked=False):
    """per-instance tolerance from the smallest singular value of the geometric Jacobian"""
    import numpy as np
    sigma_min = np.asarray(sigma_min)
    if stacked:
        return np.full(sigma_min.shape, PINV_RTOL)
    return np.where(sigma_min >= SIGMA_WELL, PINV_RTOL_WELL, PINV_RTOL)


PINV_RTOL_TIGHT = 1e-10
QP_RTOL = 1e-8
KKT_TOL = 1e-8
