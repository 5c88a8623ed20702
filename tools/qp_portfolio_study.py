#!/usr/bin/env python3
"""Can four lanes per instance shorten the QP tick of BASELINE config 4?  The tick of a batch is
its slowest instance, i.e. the worst pass count of the primal active set after the Gauss-Seidel start
(clik_qp_static.hpp::qp_box_pas); this study counts passes in numpy on the bench inputs for every portfolio the lanes
of a quad could run with ONE instruction stream: release policies (worst / all wrong multipliers / block pivoting),
sweep counts, sweep orders, over-relaxation factors - and the per-instance minimum over four of them.
    python tools/qp_portfolio_study.py [instances=16384]          (summary: profiles/r3_qp_portfolio_study.md)
"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.qp_pass_study import box_qps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
P, g, lb, ub = box_qps(B)
n = 7
idx = np.arange(n)

def gs(P, g, lb, ub, sweeps, order=None, omega=1.0, x0=None):
    order = range(n) if order is None else order
    ip = 1.0 / P[:, idx, idx]
    x = np.clip(g * ip, lb, ub) if x0 is None else x0.copy()
    res = g - np.einsum('bij,bj->bi', P, x)
    for s in range(sweeps):
        for a in order:
            xa = np.clip(x[:, a] + omega * res[:, a] * ip[:, a], lb[:, a], ub[:, a])
            dl = xa - x[:, a]
            x[:, a] = xa
            res -= P[:, :, a] * dl[:, None]
    return x

def face_solve(P, g, lb, ub, x, W):
    """minimiser on the face: held states keep x, free solve"""
    M = P.copy()
    M[:, idx, idx] += np.where(W, 1e30, 0.0)
    gr = np.einsum('bij,bj->bi', P, x) - g
    d = np.where(W, 0.0, np.linalg.solve(M, np.where(W, 0.0, gr)[..., None])[..., 0])
    return d, gr

def pas(P, g, lb, ub, x, W, policy="worst", max_it=60):
    """returns passes per instance.  policy: worst | all | bpp"""
    B = len(g)
    x = x.copy(); W = W.copy()
    done = np.zeros(B, bool); its = np.zeros(B, int)
    tol = 1e-9 * np.maximum(1.0, np.abs(g))
    for _ in range(max_it):
        if done.all(): break
        live = ~done
        its[live] += 1
        d, gr = face_solve(P, g, lb, ub, x, W)
        if policy == "bpp":
            xn = x - d
            low, high = xn < lb, xn > ub
            xn = np.clip(xn, lb, ub)
            Wn = W | low | high
            grn = np.einsum('bij,bj->bi', P, xn) - g
            push = np.where(Wn & (xn <= lb), -grn, np.where(Wn & (xn >= ub), grn, -np.inf))
            wrong = push > tol
            feas = ~(low | high).any(axis=1)
            rel = wrong & feas[:, None]          # release only when the step was feasible? (variant) -> all wrong
            rel = wrong
            fin = feas & ~wrong.any(axis=1)
            x = np.where(live[:, None], xn, x); W = np.where(live[:, None], Wn & ~rel, W)
            done = done | (live & fin)
            continue
        room = np.where(d > 0, x - lb, x - ub)
        with np.errstate(divide='ignore', invalid='ignore'):
            hit = np.maximum(np.where(d != 0, room / d, np.inf), 0.0)
        alpha = np.minimum(1.0, hit.min(axis=1))
        blocked = alpha < 1.0
        lands = blocked[:, None] & (hit <= alpha[:, None] * (1 + 1e-7)) & (d != 0)
        xn = np.where(lands, np.where(d > 0, lb, ub), x - alpha[:, None] * d)
        grn = np.einsum('bij,bj->bi', P, xn) - g
        Wn = W | lands
        push = np.where(Wn & (xn <= lb), -grn, np.where(Wn & (xn >= ub), grn, -np.inf))
        push = np.where(push > tol, push, -np.inf)
        full = ~blocked
        wrong = np.isfinite(push.max(axis=1))
        if policy == "all":
            rel = full[:, None] & np.isfinite(push)
        else:
            rel = np.zeros_like(W); sel = full & wrong
            rel[sel, push.argmax(axis=1)[sel]] = True
        x = np.where(live[:, None], xn, x); W = np.where(live[:, None], Wn & ~rel, W)
        done = done | (live & full & ~wrong)
    its[~done] = 99
    return its, x

def stats(name, its):
    print("%-46s mean %.3f  p99 %d  p99.9 %d worst %d  hist %s" % (name, its.mean(), np.percentile(its, 99), np.percentile(its, 99.9), its.max(), np.bincount(np.minimum(its, 12))))

# optimal partition via long PAS
xs = gs(P, g, lb, ub, 12)
W0 = (xs <= lb) | (xs >= ub)
its0, xopt = pas(P, g, lb, ub, xs, W0, "worst")
stats("GS12 + PAS worst (current)", its0)
Wopt = (xopt <= lb + 1e-12) | (xopt >= ub - 1e-12)
ham = (W0 != Wopt).sum(axis=1)
print("hamming(GS12 partition, optimal):", np.bincount(ham))
print("free at optimum:", np.bincount((~Wopt).sum(axis=1)))
cond = np.linalg.cond(P)
print("cond(P) pct 50/99/max: %.2e %.2e %.2e" % (np.percentile(cond, 50), np.percentile(cond, 99), cond.max()))
res = {}
for sw in (4, 6, 8, 12, 16, 24):
    xs = gs(P, g, lb, ub, sw); W = (xs <= lb) | (xs >= ub)
    for pol in ("worst", "all", "bpp"):
        its, _ = pas(P, g, lb, ub, xs, W, pol)
        res[(sw, pol)] = its
        stats("GS%d + %s" % (sw, pol), its)
for sw in (6, 12):
    m = np.minimum.reduce([res[(sw, p)] for p in ("worst", "all", "bpp")])
    stats("GS%d portfolio policies(worst,all,bpp)" % sw, m)
# orderings / relaxation portfolios with 12 sweeps
variants = {"fwd": dict(order=list(range(7))), "rev": dict(order=list(range(6, -1, -1))),
            "sor1.5": dict(omega=1.5), "sor1.8": dict(omega=1.8), "sym": None}
vr = {}
for k, kw in variants.items():
    if k == "sym":
        x = None
        for s in range(6):
            x = gs(P, g, lb, ub, 1, order=list(range(7)), x0=x)
            x = gs(P, g, lb, ub, 1, order=list(range(6, -1, -1)), x0=x)
        xs = x
    else:
        xs = gs(P, g, lb, ub, 12, **kw)
    W = (xs <= lb) | (xs >= ub)
    vr[k], _ = pas(P, g, lb, ub, xs, W, "worst")
    stats("GS12[%s] + worst" % k, vr[k])
stats("portfolio fwd,rev,sor1.5,sym", np.minimum.reduce([vr[k] for k in ("fwd", "rev", "sor1.5", "sym")]))
stats("portfolio fwd+bpp12,rev,sor1.5", np.minimum.reduce([vr["fwd"], res[(12, "bpp")], vr["rev"], vr["sor1.5"]]))

print("---- omega portfolios (same instruction stream, omega as per-lane data)")
om = {}
for w in (0.8, 1.0, 1.15, 1.3, 1.4, 1.5, 1.6, 1.7, 1.8):
    xs = gs(P, g, lb, ub, 12, omega=w); W = (xs <= lb) | (xs >= ub)
    om[w], _ = pas(P, g, lb, ub, xs, W, "worst")
    stats("GS12 omega %.2f" % w, om[w])
import itertools
best = []
for combo in itertools.combinations(sorted(om), 4):
    m = np.minimum.reduce([om[k] for k in combo])
    best.append((m.max(), (m >= 3).sum(), m.mean(), combo))
best.sort()
for b in best[:8]:
    print("portfolio", b[3], "worst %d  n(>=3) %d mean %.4f" % (b[0], b[1], b[2]))
for sw in (6, 8):
    o2 = {}
    for w in (1.0, 1.3, 1.5, 1.7):
        xs = gs(P, g, lb, ub, sw, omega=w); W = (xs <= lb) | (xs >= ub)
        o2[w], _ = pas(P, g, lb, ub, xs, W, "worst")
    stats("GS%d portfolio omega(1,1.3,1.5,1.7)" % sw, np.minimum.reduce(list(o2.values())))
