#!/usr/bin/env python3
"""numpy prototype of the box-QP active set on a SWEPT tableau (what qp_box_pas runs since round 4): the symmetric sweep
operator keeps S with  S_FF = -(P_FF)^-1  for the free set F, one sweep (a rank-one update, one reciprocal) per state that
changes sides, instead of a masked 7 x 7 LDL' refactorisation per pass.  Checks the minimisers against a brute-force
KKT solve and counts passes / sweeps on the reduced QPs of BASELINE config 4.
    python tools/qp_sweep_proto.py [instances=4096]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from qp_pass_study import box_qps                   # noqa: E402


def sweep(S, k, sign):
    """symmetric sweep of index k in place; sign = +1 forward (k joins the swept set), -1 reverse"""
    d = S[k, k]
    col = S[:, k].copy()
    S -= np.outer(col, col) / d
    S[:, k] = sign * col / d
    S[k, :] = S[:, k]
    S[k, k] = -1.0 / d


def solve_one(P, g, lb, ub, n_sweeps=12, max_pass=40):
    n = len(g)
    ip = 1.0 / np.diag(P)
    x = np.clip(g * ip, lb, ub)
    res = g - P.dot(x)
    for _ in range(n_sweeps):
        for a in range(n):
            xa = min(max(x[a] + res[a] * ip[a], lb[a]), ub[a])
            dl = xa - x[a]
            x[a] = xa
            res -= P[:, a] * dl
    held = (x <= lb) | (x >= ub)
    x = np.where(x <= lb, lb, np.where(x >= ub, ub, x))
    S = P.copy()
    n_sw = 0
    for k in range(n):
        if not held[k]:
            sweep(S, k, +1.0)
            n_sw += 1
    gr = P.dot(x) - g
    tol = 1e-9 * np.maximum(1.0, np.abs(g))
    for p in range(max_pass):
        free = ~held
        d = np.where(free, -S.dot(np.where(free, gr, 0.0)), 0.0)          # (P_FF)^-1 gr_F
        tgt = np.where(d > 0, lb, ub)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(d != 0, np.abs(x - tgt) / np.abs(d), np.inf)
        amin = min(1.0, r.min())
        blocked = amin < 1.0
        lands = blocked & (r <= amin * (1 + 1e-7))
        x = np.where(lands, tgt, x - amin * d)
        for k in np.nonzero(lands)[0]:
            sweep(S, k, -1.0)
            n_sw += 1
        held = held | lands
        gr = P.dot(x) - g
        if blocked:
            continue
        push = np.where(x <= lb, -gr, gr)
        c = np.where(held & (ub > lb), push, -np.inf) - tol
        if c.max() > 0:
            k = int(np.argmax(c))
            held[k] = False
            sweep(S, k, +1.0)
            n_sw += 1
            continue
        # one refinement step on the final face (the rank-one updates accumulate rounding; Newton corrects itself)
        free = ~held
        x = x - np.where(free, -S.dot(np.where(free, gr, 0.0)), 0.0)
        return x, p + 1, n_sw
    raise RuntimeError("pass cap")


def brute(P, g, lb, ub):
    """exact minimiser by enumerating the 3^n partitions is too slow for n = 7: projected Newton to machine precision"""
    from scipy.optimize import minimize
    n = len(g)
    r = minimize(lambda v: 0.5 * v.dot(P).dot(v) - g.dot(v), np.clip(np.linalg.solve(P, g), lb, ub),
                 jac=lambda v: P.dot(v) - g, bounds=list(zip(lb, ub)), method="L-BFGS-B",
                 options=dict(ftol=1e-30, gtol=1e-14, maxiter=5000, maxfun=50000))
    return r.x


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    P, g, lb, ub = box_qps(B)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import clik_oracle as orc
    passes, sweeps, worst = [], [], 0.0
    for b in range(B):
        x, p, s = solve_one(P[b], g[b].copy(), lb[b], ub[b])
        passes.append(p)
        sweeps.append(s)
        # KKT check of the returned point
        gr = P[b].dot(x) - g[b]
        tol = 1e-9 * np.maximum(1.0, np.abs(g[b]))
        ok = (((x <= lb[b]) & (gr >= -tol)) | ((x >= ub[b]) & (gr <= tol)) | (np.abs(gr) <= tol)).all()
        assert ok and (x >= lb[b] - 1e-15).all() and (x <= ub[b] + 1e-15).all(), (b, gr, x)
        if b < 256:
            xo = orc.qp_solve_dense(np.ones(len(x)), np.eye(len(x)), lb[b], ub[b]) if False else None
        worst = max(worst, float(np.abs(gr[(x > lb[b]) & (x < ub[b])]).max(initial=0.0)))
    passes, sweeps = np.array(passes), np.array(sweeps)
    print("instances %d: passes mean %.3f p99.9 %d worst %d (histogram %s); sweeps mean %.2f worst %d; free-state gradient "
          "at the answer <= %.1e" % (B, passes.mean(), np.percentile(passes, 99.9), passes.max(),
                                     np.bincount(passes).tolist(), sweeps.mean(), sweeps.max(), worst))


if __name__ == "__main__":
    main()
