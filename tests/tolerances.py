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
PINV_RTOL_TIGHT = 1e-10
QP_RTOL = 1e-8
KKT_TOL = 1e-8
