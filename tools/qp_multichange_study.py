#!/usr/bin/env python3
"""Companion of qp_portfolio_study.py: do passes that change SEVERAL states at once (clamp every violator of the face
minimiser, release every wrong multiplier; k block-pivot passes before the single-change active set) shorten the worst
instance of BASELINE config 4?  (No: profiles/r3_qp_portfolio_study.md.)
    python tools/qp_multichange_study.py [instances=16384] [seed=0]
"""
import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
from tools.qp_pass_study import box_qps
import importlib.util
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
try:
    P, g, lb, ub = box_qps(B, seed=seed)
except TypeError:
    P, g, lb, ub = box_qps(B)
n = 7; idx = np.arange(n)
def gs(P, g, lb, ub, sweeps, omega=1.0):
    ip = 1.0 / P[:, idx, idx]
    x = np.clip(g * ip, lb, ub)
    res = g - np.einsum('bij,bj->bi', P, x)
    for s in range(sweeps):
        for a in range(n):
            xa = np.clip(x[:, a] + omega * res[:, a] * ip[:, a], lb[:, a], ub[:, a])
            dl = xa - x[:, a]; x[:, a] = xa; res -= P[:, :, a] * dl[:, None]
    return x
def face_solve(P, g, x, W):
    M = P.copy(); M[:, idx, idx] += np.where(W, 1e30, 0.0)
    gr = np.einsum('bij,bj->bi', P, x) - g
    d = np.where(W, 0.0, np.linalg.solve(M, np.where(W, 0.0, gr)[..., None])[..., 0])
    return d, gr
def run(P, g, lb, ub, x, W, kbpp=0, multiadd=False, relall=False, max_it=40):
    B = len(g); x = x.copy(); W = W.copy()
    done = np.zeros(B, bool); its = np.zeros(B, int)
    tol = 1e-9 * np.maximum(1.0, np.abs(g))
    for it in range(max_it):
        if done.all(): break
        live = ~done; its[live] += 1
        d, gr = face_solve(P, g, x, W)
        xfull = x - d
        viol = (xfull < lb - 1e-14) | (xfull > ub + 1e-14)
        if it < kbpp or multiadd:
            # clamp all violators
            xn = np.clip(xfull, lb, ub); lands = viol
            blocked = viol.any(axis=1)
        else:
            room = np.where(d > 0, x - lb, x - ub)
            with np.errstate(divide='ignore', invalid='ignore'):
                hit = np.maximum(np.where(d != 0, room / d, np.inf), 0.0)
            alpha = np.minimum(1.0, hit.min(axis=1)); blocked = alpha < 1.0
            lands = blocked[:, None] & (hit <= alpha[:, None] * (1 + 1e-7)) & (d != 0)
            xn = np.where(lands, np.where(d > 0, lb, ub), x - alpha[:, None] * d)
        grn = np.einsum('bij,bj->bi', P, xn) - g
        Wn = W | lands
        push = np.where(Wn & (xn <= lb), -grn, np.where(Wn & (xn >= ub), grn, -np.inf))
        push = np.where(push > tol, push, -np.inf)
        wrong = np.isfinite(push.max(axis=1))
        if it < kbpp:
            rel = np.isfinite(push)                       # BPP: release all wrong, even when blocked
            fin = ~blocked & ~wrong
        else:
            full = ~blocked
            if relall: rel = full[:, None] & np.isfinite(push)
            else:
                rel = np.zeros_like(W); sel = full & wrong
                rel[sel, push.argmax(axis=1)[sel]] = True
            fin = full & ~wrong
        x = np.where(live[:, None], xn, x); W = np.where(live[:, None], Wn & ~rel, W)
        done = done | (live & fin)
    its[~done] = 99
    return its, x
def stats(name, its):
    print("%-40s mean %.3f p99.9 %d worst %d hist %s" % (name, its.mean(), np.percentile(its, 99.9), its.max(), np.bincount(np.minimum(its, 12))), flush=True)
for sw in (12, 8, 6):
    xs = gs(P, g, lb, ub, sw); W0 = (xs <= lb) | (xs >= ub)
    its, xo = run(P, g, lb, ub, xs, W0); stats("GS%d + PAS worst" % sw, its)
    f0 = 0.5*np.einsum('bi,bij,bj->b', xo, P, xo) - (g*xo).sum(1)
    for k in (1, 2, 3, 4):
        its, x = run(P, g, lb, ub, xs, W0, kbpp=k); stats("GS%d + bpp%d then worst" % (sw, k), its)
        f = 0.5*np.einsum('bi,bij,bj->b', x, P, x) - (g*x).sum(1); print("   max obj diff", np.abs(f-f0).max())
    its, x = run(P, g, lb, ub, xs, W0, multiadd=True); stats("GS%d + multiadd/worst" % sw, its)
    its, x = run(P, g, lb, ub, xs, W0, multiadd=True, relall=True); stats("GS%d + multiadd/all" % sw, its)
