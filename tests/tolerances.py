"""The stated parity tolerance (fp64) - ONE conditioning-aware rule, used by the tests, by `smoke()` and by every
tools/fuzz_*.py (no tool may accept an error the stated rule rejects).

What is compared: per robot instance  err = max_j |v_hip - v_oracle| / (1 + max_j |v_oracle|)  over the velocities (and
slacks) of one tick.

The rule:       err  <=  max(FLOOR, FACTOR * u * kappa),       u = 2^-53,  FACTOR = 8,  FLOOR = 1e-12

kappa is the condition number of what the REFERENCE's algorithm hands to its linear solver for that instance - its own
answer is defined only up to c * u * kappa, whichever correct solver (CasADi's, numpy's LU, the kernels' LDL^T) runs:

  pinv   the SUM of the 2-norm condition numbers of the symmetric matrices `J J' + lam I` / `J' J + lam I` of every
         pseudo-inverse the accepted mode evaluates (pseudo_inverse.py:92-105; the rounding errors of successive solves
         add up: a six-constraint stack projects five times), collected by the oracle itself
         (`clik_oracle.pinv_solve_batch(..., cond_out=)`).  Closed forms per solve, for batches too large for the numpy
         oracle:
           single task, m <= n rows          kappa = (smax^2 + lam) / (smin^2 + lam)
           an equality behind the first one  kappa = (c smax^2 + lam) / lam: the reference projects through the doubly
                                             stacked [J; J] (:317-326 + :382-396; c = 2, plus the rows stacked with
                                             it), an n x n Gram matrix of rank < n for EVERY configuration
         At the default damping 1e-7 on the vendored arms (smax^2 <= 5.7) the stacked form gives <= 1e-7 = PINV_RTOL, the
         flat bound earlier rounds stated; it scales as 1 / lam (lam = 1e-9: 1e-5, measured 2.3e-7).
  QP     cond(H) * cond+(Aa H^-1 Aa') of the rows active at the minimiser (`clik_oracle.qp_condition`).  A STATUS may
         differ from the oracle's only where the rows' LP feasibility margin is within LP_MARGIN of zero ("infeasible"
         and "solved to tolerance" are both defensible there), or where the oracle's own active-set method gave up and
         the device's answer passes the KKT check at KKT_TOL.

Instances whose bound exceeds ILL_POSED (undamped inverse of a rank-deficient stack, lam ~ 1e-16) are no parity
evidence either way and are left out, counted (`LAST["left_out"]`); a comparison that leaves out more than MAX_LEFT_OUT
of its instances - or all of them - FAILS: it compared nothing.  A NaN from the device where the reference is finite
fails.  Modes must be identical wherever the smallest tangent-cone decision
margin of the instance (`clik_oracle.tangent_cone_margin`) exceeds MODE_MARGIN.

Measured against the reference's own Python over the stand-in casadi (tests/golden/ref_pins.npz): err / (u kappa) <= 0.8
for the oracle (tests/test_refpins.py); for the HIP kernels - different algebra: LDL' without pivoting, push-through and
Woodbury forms, DESIGN.md section 3 - <= FACTOR with the margins recorded in profiles/r4_tolerance_sweep.txt."""
import numpy as np

U = 2.0 ** -53
FACTOR = 8.0
FLOOR = 1e-12
ILL_POSED = 1e-3
MODE_MARGIN = 1e-7
LP_MARGIN = 1e-6
KKT_TOL = 1e-8
# ceilings of the rule at the DEFAULT options on the vendored arms (for quick checks on a handful of instances where no
# per-instance kappa is at hand; never looser than the rule's own worst case there):
PINV_RTOL = 1e-7          # = FACTOR u (2 smax^2 + lam) / lam at lam = 1e-7, smax^2 = 5.6
QP_RTOL = 1e-8            # = FACTOR u kappa at kappa = 1.1e7 (cond(H) = 1001 with unit weights, mu = 1e-3)
PINV_RTOL_TIGHT = 1e-10   # single well-conditioned task (kappa < 1e5)


def rtol_from_cond(kappa):
    """the rule: per-instance bound from the condition number(s) the oracle reported"""
    return np.maximum(FLOOR, FACTOR * U * np.asarray(kappa, dtype=float))


def kappa_single(sigma_min, sigma_max, lam):
    sigma_min, sigma_max = np.asarray(sigma_min, dtype=float), np.asarray(sigma_max, dtype=float)
    return (sigma_max ** 2 + lam) / (sigma_min ** 2 + lam)


def kappa_stacked(sigma_max, lam, copies=2.0):
    return (copies * np.asarray(sigma_max, dtype=float) ** 2 + lam) / lam


def pinv_rtol(sigma_min, stacked=False, sigma_max=None, lam=1e-7):
    """closed-form bound from the singular values of the first task's Jacobian (see the module text); `sigma_max`
    defaults to the vendored arms' largest (2.4)"""
    sigma_min = np.asarray(sigma_min, dtype=float)
    smax = np.full(sigma_min.shape, 2.4) if sigma_max is None else np.asarray(sigma_max, dtype=float)
    kappa = kappa_stacked(smax, lam) if stacked else kappa_single(sigma_min, smax, lam)
    return rtol_from_cond(kappa)


def rel_err(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))


def check_pinv(dq, ref, kappa, what=""):
    """assert the rule on a batch; returns (worst err, worst err / (u kappa), instances left out as ill-posed)"""
    err = rel_err(dq, ref)
    tol = rtol_from_cond(kappa)
    posed = tol < ILL_POSED
    bad = posed & (err > tol)
    assert not bad.any(), (what, "instances beyond the stated tolerance", int(bad.sum()), float((err / tol)[posed].max()),
                           float(err[bad].max()))
    ratio = float((err[posed] / (U * np.asarray(kappa)[posed])).max()) if posed.any() else 0.0
    return (float(err[posed].max()) if posed.any() else 0.0), ratio, int((~posed).sum())


def _kappa_for(ref):
    from oracle import clik_oracle
    return clik_oracle.condition_of(ref)


MAX_LEFT_OUT = 0.25     # a comparison that leaves more than this share of its instances out as ill-posed is no comparison
LAST = {}               # what the most recent *_close / worst_over_tol call did: {"rule", "checked", "left_out", "worst"}


def _compare(a, ref, rows, kappa, ceiling, what):
    """the shared body of pinv_close / qp_close.  `ref` rows that are NaN (the oracle's "infeasible") are no instances;
    a NaN in `a` where `ref` is finite is a FAILURE.  `rows`: boolean mask of the instances to compare (so that `ref` can
    stay the array the oracle returned and every instance keeps ITS kappa); `kappa`: condition numbers [B] given
    explicitly (`cond_out` of the oracle), else looked up from the oracle's last solve when `ref` is its result."""
    a, ref = np.atleast_2d(np.asarray(a, dtype=float)), np.atleast_2d(np.asarray(ref, dtype=float))
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    keep = ~np.isnan(ref).any(axis=1)
    if rows is not None:
        keep &= np.asarray(rows, dtype=bool)
    if kappa is None:
        kappa = _kappa_for(ref)
    LAST.clear()
    if np.isnan(a[keep]).any():
        LAST.update(rule="nan", checked=int(keep.sum()), left_out=0, worst=float("inf"))
        return False
    err = np.zeros(len(ref))
    err[keep] = rel_err(a[keep], ref[keep])
    if kappa is None:
        LAST.update(rule="ceiling", checked=int(keep.sum()), left_out=0, worst=float(err.max() / ceiling) if len(err) else 0.0)
        return bool((err[keep] < ceiling).all())
    tol = rtol_from_cond(np.asarray(kappa, dtype=float))
    assert tol.shape == err.shape, (what, "kappa does not belong to this batch", tol.shape, err.shape)
    posed = keep & (tol < ILL_POSED)
    left_out = int((keep & ~posed).sum())
    LAST.update(rule="kappa", checked=int(posed.sum()), left_out=left_out,
                worst=float((err[posed] / tol[posed]).max()) if posed.any() else 0.0)
    if keep.any() and (not posed.any() or left_out > MAX_LEFT_OUT * keep.sum()):
        return False            # (nothing, or too little, was actually compared)
    return bool((err[posed] <= tol[posed]).all())


def pinv_close(a, ref, ceiling=PINV_RTOL, rows=None, kappa=None):
    """`a` against `ref` under the rule.  With `kappa` [B] given, or when `ref` is what the numpy oracle's last solve
    returned (the array itself or a basic slice of it), every instance is held to ITS bound max(FLOOR, FACTOR u kappa);
    instances beyond ILL_POSED are left out, COUNTED in `LAST`, and the comparison FAILS when it leaves out more than
    MAX_LEFT_OUT of them (or all).  Otherwise - a comparison of two device results, a fixture - the default-options
    ceiling applies to every row.  Select instances with `rows=mask`, not by indexing `ref` (a fancy-indexed copy is
    no longer the oracle's array: it would silently fall back to the ceiling).  A NaN from the device where the
    reference is finite fails."""
    return _compare(a, ref, rows, kappa, ceiling, "pinv_close")


def qp_close(a, ref, ceiling=QP_RTOL, rows=None, kappa=None):
    """the same for the QP path (kappa = clik_oracle.qp_condition of each instance; rows the oracle reports infeasible -
    NaN in `ref` - are no instances)"""
    return _compare(a, ref, rows, kappa, ceiling, "qp_close")


def worst_over_tol(a, ref, rows=None, kappa=None):
    """max over the well-posed instances of err / (the rule's bound), kappa given or looked up from the oracle solve that
    returned `ref` (then `ref` must be that array or a basic slice of it; `rows`: boolean mask of the instances to look
    at).  The sweeps of tools/fuzz_*.py flag a skill when this exceeds 1.  A NaN from the device on an instance the oracle
    solved counts as an infinite error.  Returns (ratio, worst err, instances left out as ill-posed)."""
    if kappa is None:
        kappa = _kappa_for(ref)
    assert kappa is not None, "worst_over_tol needs the array the oracle returned (or kappa=)"
    a, ref = np.atleast_2d(a), np.atleast_2d(ref)
    keep = ~np.isnan(ref).any(axis=1)
    if rows is not None:
        keep &= rows
    tol = rtol_from_cond(kappa)
    posed = keep & (tol < ILL_POSED)
    if not posed.any():
        return 0.0, 0.0, int((keep & ~posed).sum())
    if np.isnan(a[posed]).any():
        return float("inf"), float("inf"), int((keep & ~posed).sum())
    err = rel_err(a[posed], ref[posed])
    return float((err / tol[posed]).max()), float(err.max()), int((keep & ~posed).sum())
