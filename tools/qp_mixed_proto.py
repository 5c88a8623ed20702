#!/usr/bin/env python3
"""numpy prototype of the mixed primal active set (clik_qp_static.hpp::qp_mixed_pas): the QP family
   box + hard general inequality rows + soft inequality rows lifted into the box,
instance by instance against the oracle's dense Goldfarb-Idnani: the Moe-2016 wall skill (hard and soft walls; inside,
near and outside the walls) and 1500 random problems (4-8 variables, 1-4 general rows, infeasible ones included).
    python tools/qp_mixed_proto.py
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from casclik_amd import skills
from oracle import clik_oracle as orc
from extern_skills import moe_box_skill


def reduce_problem(H, A, lbA, ubA, n):
    """fold soft equalities; lift soft inequalities; returns per-instance (P, g, lb, ub, G, lbg, ubg, info)"""
    nv = H.shape[0]
    ns = nv - n
    soft_col = {}
    for r in range(A.shape[0]):
        cols = np.nonzero(A[r, n:])[0]
        if cols.size:
            soft_col[r] = n + cols[0]
    P = np.diag(H[:n]).astype(float)
    g = np.zeros(n)
    lb = np.full(n, -np.inf); ub = np.full(n, np.inf)
    hard_rows, lift = [], []
    for r in range(A.shape[0]):
        a = A[r, :n]
        if r in soft_col:
            h = H[soft_col[r]]
            if lbA[r] == ubA[r]:
                P += h * np.outer(a, a); g += h * lbA[r] * a
            else:
                lift.append((r, a, h))
        else:
            nzc = np.nonzero(a)[0]
            if nzc.size == 1 and abs(a[nzc[0]] - 1.0) < 1e-15:
                lb[nzc[0]] = max(lb[nzc[0]], lbA[r]); ub[nzc[0]] = min(ub[nzc[0]], ubA[r])
            else:
                hard_rows.append(r)
    nz = n + len(lift)
    Pz = np.zeros((nz, nz)); Pz[:n, :n] = P
    gz = np.zeros(nz); gz[:n] = g
    lbz = np.full(nz, -np.inf); ubz = np.full(nz, np.inf); lbz[:n] = lb; ubz[:n] = ub
    for k, (r, a, h) in enumerate(lift):
        Pz[:n, :n] += h * np.outer(a, a); Pz[:n, n + k] = -h * a; Pz[n + k, :n] = -h * a; Pz[n + k, n + k] = h
        lbz[n + k], ubz[n + k] = lbA[r], ubA[r]
    G = np.zeros((len(hard_rows), nz)); G[:, :n] = A[hard_rows, :n]
    return Pz, gz, lbz, ubz, G, lbA[hard_rows], ubA[hard_rows], (lift, hard_rows)


def mixed_pas(P, g, lb, ub, G, lbg, ubg, sweeps=12, max_pass=40):
    n, nh = len(g), len(lbg)
    idx = np.arange(n)
    ip = 1.0 / np.diag(P)
    x = np.clip(g * ip, lb, ub)
    res = g - P @ x
    for s in range(sweeps):
        for a in range(n):
            xa = min(max(x[a] + res[a] * ip[a], lb[a]), ub[a]); dl = xa - x[a]; x[a] = xa; res -= P[:, a] * dl
    if nh:
        # rows the box start violates: a few sweeps with those rows pulled in by a penalty (see qp_mixed_pas)
        gx = G @ x
        hi = gx > ubg + 1e-12 * np.maximum(1, np.abs(ubg))
        lo = gx < lbg - 1e-12 * np.maximum(1, np.abs(lbg))
        if (hi | lo).any():
            cf = np.where(hi | lo, 1e3 * np.diag(P).max(), 0.0)
            bv = np.where(hi, ubg, np.where(lo, lbg, 0.0))
            P2 = P + (G.T * cf) @ G
            g2 = g + G.T @ (cf * bv)
            ip2 = 1.0 / np.diag(P2)
            res = g2 - P2 @ x
            for s_ in range(8):
                for a in range(n):
                    xa = min(max(x[a] + res[a] * ip2[a], lb[a]), ub[a]); dl = xa - x[a]; x[a] = xa; res -= P2[:, a] * dl
    held = (x <= lb) | (x >= ub)
    gx = G @ x
    act = np.where(gx > ubg + 1e-12 * np.maximum(1, np.abs(ubg)), 1, np.where(gx < lbg - 1e-12 * np.maximum(1, np.abs(lbg)), -1, 0))
    tol = 1e-9 * np.maximum(1.0, np.abs(g))
    passes = 0
    for it in range(max_pass):
        passes += 1
        M = P + np.diag(np.where(held, 1e30, 0.0))
        gr = P @ x - g
        d0 = np.where(held, 0.0, np.linalg.solve(M, np.where(held, 0.0, gr)))
        if nh:
            Y = np.linalg.solve(M, G.T); Y[held, :] = 0.0
            S0 = G @ Y
            S = S0 + np.diag(np.where(act != 0, 1e-13, 1e30))
            bnd = np.where(act > 0, ubg, lbg)
            resid = np.where(act != 0, G @ x - bnd, 0.0)
            lam = np.linalg.solve(S, resid - G @ d0)
            lam = lam + np.linalg.solve(S, np.where(act != 0, 1e-13 * lam, 0.0))      # (one step of iterative refinement)
            lam = np.where(act != 0, lam, 0.0)
            d = d0 + Y @ lam
        else:
            lam = np.zeros(0); d = d0
        d = np.where(held, 0.0, d)
        with np.errstate(divide='ignore', invalid='ignore'):
            room = np.where(d > 0, x - lb, x - ub)
            hit = np.where(d != 0, np.maximum(room / d, 0.0), np.inf)
            if nh:
                sl = G @ d; gx = G @ x
                roomr = np.where(sl > 0, gx - lbg, gx - ubg)
                hitr = np.where((act == 0) & (sl != 0), np.maximum(roomr / sl, 0.0), np.inf)
            else:
                hitr = np.zeros(0)
        amin = min(1.0, hit.min() if n else 1.0, hitr.min() if nh else 1.0)
        blocked = amin < 1.0
        thr = amin * (1 + 1e-7)
        lands = blocked & (hit <= thr) & (d != 0)
        xn = np.where(lands, np.where(d > 0, lb, ub), x - amin * d)
        held = held | lands
        if nh:
            landr = blocked & (hitr <= thr)
            act = np.where(landr, np.where(sl > 0, -1, 1), act)
        x = xn
        if blocked:
            continue
        # face minimum: multipliers
        gfull = P @ x - g + (G.T @ lam if nh else 0.0)
        push = np.where(held & (x <= lb), -gfull, np.where(held & (x >= ub), gfull, -np.inf))
        push = np.where(held & (ub > lb), push, -np.inf) - tol
        worst, wi, wkind = 0.0, -1, 0
        if push.max() > 0:
            worst, wi, wkind = push.max(), int(push.argmax()), 1
        if nh:
            pr = np.where(act != 0, -act * lam, -np.inf) - 1e-9 * np.maximum(1.0, np.abs(lam))
            if pr.max() > 0 and pr.max() > worst:
                worst, wi, wkind = pr.max(), int(pr.argmax()), 2
        if wkind == 0:
            if nh:
                mag = np.maximum(1.0, np.abs(G * x[None, :]).max(axis=1))
                bnd_ = np.where(act > 0, ubg, lbg)
                dev = np.where(act != 0, np.abs(G @ x - bnd_), 0.0)
                sc = np.maximum(mag, np.abs(bnd_))
                if np.any(dev > 1e-8 * sc):
                    return x, lam, 2, passes       # an active row cannot be met on any face reachable: infeasible
            return x, lam, 0, passes
        if wkind == 1:
            held[wi] = False
        else:
            act[wi] = 0
    return x, lam, 1, passes


def run(spec, n, t, Q, Y=None, label=""):
    H, A, lbA, ubA = orc.qp_data_batch(spec, t, Q, Y=Y)
    stat = {"ok": 0, "cap": 0, "mismatch": 0, "infeas_ref": 0}
    worst = 0.0; pas_hist = []
    for b in range(len(Q)):
        try:
            xr = orc.qp_solve_dense(H[b], A[b], lbA[b], ubA[b])
        except orc.QPInfeasible:
            stat["infeas_ref"] += 1; continue
        Pz, gz, lbz, ubz, G, lbg, ubg, info = reduce_problem(H[b], A[b], lbA[b], ubA[b], n)
        if (lbz > ubz).any():
            continue
        x, lam, st, passes = mixed_pas(Pz, gz, lbz, ubz, G, lbg, ubg)
        pas_hist.append(passes)
        if st != 0:
            stat["cap"] += 1; continue
        err = np.abs(x[:n] - xr[:n]).max() / (1 + np.abs(xr[:n]).max())
        worst = max(worst, err)
        if err > 1e-8: stat["mismatch"] += 1
        else: stat["ok"] += 1
    print(label, stat, "worst rel err %.2e" % worst, "passes hist", np.bincount(pas_hist))


def moe_test():
    ur5 = skills.ur5()
    rng = np.random.default_rng(0)
    B = 400
    for soft in (False, True):
        spec, home = moe_box_skill(ur5, soft_walls=soft)
        for name, scale in (("inside", 0.03), ("near", 0.10), ("outside", 0.35)):
            Q = home + rng.normal(scale=scale, size=(B, 6))
            run(spec, 6, 3.0, Q, label="moe %s %s" % ("soft" if soft else "hard", name))


def random_test(nprob=3000, seed=1):
    rng = np.random.default_rng(seed)
    stat = {"ok": 0, "cap": 0, "mismatch": 0, "infeas_ref": 0, "cap_on_infeasible": 0, "ok_but_ref_infeasible": 0}
    hist = []; worst = 0.0
    for k in range(nprob):
        n = int(rng.integers(4, 9)); nh = int(rng.integers(1, 5))
        J = rng.normal(size=(6, n)) * rng.uniform(0.1, 1.0, size=(6, 1))
        P = 1e-3 * np.eye(n) + 1.001 * J.T @ J
        v0 = rng.normal(size=n) * 2.0
        g = P @ v0
        w = rng.uniform(0.2, 2.0, n)
        lb, ub = -w, w
        free_b = rng.random(n) < 0.2
        lb = np.where(free_b, -np.inf, lb); ub = np.where(free_b, np.inf, ub)
        G = rng.normal(size=(nh, n))
        c = G @ np.clip(v0, -1, 1) + rng.normal(size=nh) * 0.5
        wr = rng.uniform(0.05, 1.0, nh)
        lbg, ubg = c - wr, c + wr
        one = rng.random(nh) < 0.3
        ubg = np.where(one, 1e10, ubg)
        # reference through the transformed problem
        L = np.linalg.cholesky(P)
        Linv = np.linalg.inv(L)
        rows = [np.eye(n)[i] for i in range(n) if np.isfinite(lb[i])] + list(G)
        lo = [lb[i] for i in range(n) if np.isfinite(lb[i])] + list(lbg)
        hi = [ub[i] for i in range(n) if np.isfinite(lb[i])] + list(ubg)
        Aall = np.array(rows)
        At = Aall @ Linv.T
        sh = Aall @ v0
        try:
            u = orc.qp_solve_dense(np.ones(n), At, np.array(lo) - sh, np.array(hi) - sh)
            xr = v0 + Linv.T @ u; ref_ok = True
        except orc.QPInfeasible:
            ref_ok = False
        x, lam, st, passes = mixed_pas(P, g, lb, ub, G, lbg, ubg)
        if not ref_ok:
            stat["infeas_ref"] += 1
            if st != 0: stat["cap_on_infeasible"] += 1
            else:
                viol = max((G @ x - ubg).max(), (lbg - G @ x).max())
                if viol < 1e-7: stat["ok_but_ref_infeasible"] += 1
                else: stat["cap_on_infeasible"] += 1
            continue
        hist.append(passes)
        if st != 0:
            stat["cap"] += 1; continue
        err = np.abs(x - xr).max() / (1 + np.abs(xr).max())
        if err > 1e-8: stat["mismatch"] += 1
        else: stat["ok"] += 1; worst = max(worst, err)
    print("random", stat, "worst rel err %.2e" % worst, "passes hist", np.bincount(hist))


if __name__ == "__main__":
    moe_test()
    random_test(1500)
