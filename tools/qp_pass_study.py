#!/usr/bin/env python3
"""Pass counts of the primal active-set iteration (qp_box_pas, casclik_amd/csrc/clik_qp_static.hpp) on the reduced box
QPs of BASELINE config 4, in numpy: which start and which release rule the kernel should use.  The tick of a batch is
its SLOWEST instance (every wave has a SIMD to itself), so the worst count matters, not the mean.

    python tools/qp_pass_study.py [instances=16384] [--portfolio]

Prints, per start (vertex the linear term points to / clipped unconstrained minimiser / clipped coordinate-wise
minimisers = what the kernel uses / box centre) and release rule (worst wrong multiplier / all wrong multipliers):
mean, 99th percentile and worst pass count; --portfolio also the per-instance minimum over every four of them."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casclik_amd import skills                      # noqa: E402
from oracle import clik_oracle as orc               # noqa: E402  (tools may use the checker)


def box_qps(B, seed=0):
    """P, g, lb, ub of  min 1/2 v'P v - g'v, lb <= v <= ub  after folding the soft pose rows (clik_qp_static.hpp)"""
    fk = skills.iiwa()
    spec = skills.qp_skill(fk)
    Q, Y = skills.synthetic_inputs(fk, B, seed=seed, distribution="mixed")
    H, A, lbA, ubA = orc.qp_data_batch(spec, 0.0, Q, Y=Y)
    n = Q.shape[1]
    soft = np.abs(A[0][:, n:]).sum(axis=1) > 0
    J, b, hs = A[:, soft, :n], lbA[:, soft], H[:, n:]
    P = np.einsum('bij,bi,bik->bjk', J, hs, J) + np.einsum('bj,jk->bjk', H[:, :n], np.eye(n))
    g = np.einsum('bij,bi->bj', J, hs * b)
    lb, ub = np.full((B, n), -np.inf), np.full((B, n), np.inf)
    for r in np.nonzero(~soft)[0]:
        col = int(np.argmax(np.abs(A[0, r, :n])))
        lb[:, col] = np.maximum(lb[:, col], lbA[:, r])
        ub[:, col] = np.minimum(ub[:, col], ubA[:, r])
    return P, g, lb, ub


def passes(P, g, lb, ub, start, release_all, max_it=80):
    B, n = g.shape
    idx = np.arange(n)
    if start == "vertex":
        x = np.where(g > 0, ub, np.where(g < 0, lb, np.clip(0.0, lb, ub)))
        W = np.ones((B, n), bool)
    elif start == "clipped":
        x = np.clip(np.linalg.solve(P, g[..., None])[..., 0], lb, ub)
        W = (x <= lb) | (x >= ub)
    elif start == "coordinate":
        x = np.clip(g / P[:, idx, idx], lb, ub)
        W = (x <= lb) | (x >= ub)
    else:
        x = np.clip(0.0, lb, ub) + 0 * g
        W = np.zeros((B, n), bool)
    gr = np.einsum('bij,bj->bi', P, x) - g
    done, its = np.zeros(B, bool), np.zeros(B, int)
    for _ in range(max_it):
        if done.all():
            break
        live = ~done
        its[live] += 1
        M = P.copy()
        M[:, idx, idx] += np.where(W, 1e30, 0.0)
        d = np.where(W, 0.0, np.linalg.solve(M, np.where(W, 0.0, gr)[..., None])[..., 0])
        room = np.where(d > 0, x - lb, x - ub)
        with np.errstate(divide='ignore', invalid='ignore'):
            hit = np.maximum(np.where(d != 0, room / d, np.inf), 0.0)
        alpha = np.minimum(1.0, hit.min(axis=1))
        blocked = alpha < 1.0
        lands = blocked[:, None] & (hit <= alpha[:, None] * (1 + 1e-7)) & (d != 0)
        xn = np.where(lands, np.where(d > 0, lb, ub), x - alpha[:, None] * d)
        grn = np.einsum('bij,bj->bi', P, xn) - g
        Wn = W | lands
        tol = 1e-9 * np.maximum(1.0, np.abs(g))
        push = np.where(Wn & (xn <= lb), -grn, np.where(Wn & (xn >= ub), grn, -np.inf))
        push = np.where(push > tol, push, -np.inf)
        full = ~blocked
        wrong = np.isfinite(push.max(axis=1))
        if release_all:
            rel = full[:, None] & np.isfinite(push)
        else:
            rel = np.zeros_like(W)
            sel = full & wrong
            rel[sel, push.argmax(axis=1)[sel]] = True
        x = np.where(live[:, None], xn, x)
        gr = np.where(live[:, None], grn, gr)
        W = np.where(live[:, None], Wn & ~rel, W)
        done = done | (live & full & ~wrong)
    grf = np.einsum('bij,bj->bi', P, x) - g
    tol = 1e-7 * np.maximum(1.0, np.abs(g))
    kkt = (((x <= lb) & (grf >= -tol)) | ((x >= ub) & (grf <= tol)) | (np.abs(grf) <= tol)).all(axis=1)
    assert done.all() and kkt.all()
    return its


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
    P, g, lb, ub = box_qps(B)
    free = ((np.linalg.solve(P, g[..., None])[..., 0] > lb) & (np.linalg.solve(P, g[..., None])[..., 0] < ub)).sum(axis=1)
    print("instances %d; free states of the clipped unconstrained minimiser (histogram 0..7): %s" % (B, np.bincount(free, minlength=8)))
    res = {}
    for start in ("vertex", "clipped", "coordinate", "centre"):
        for rel in (False, True):
            its = res[(start, "all" if rel else "worst")] = passes(P, g, lb, ub, start, rel)
            print("%-10s release %-5s  mean %.2f  p99 %d  worst %d" % (start, "all" if rel else "worst", its.mean(),
                                                                      np.percentile(its, 99), its.max()))
    if "--portfolio" in sys.argv:
        for combo in itertools.combinations(list(res), 4):
            m = np.min([res[k] for k in combo], axis=0)
            print("min of", combo, " mean %.2f  p99 %d  worst %d" % (m.mean(), np.percentile(m, 99), m.max()))


if __name__ == "__main__":
    main()
