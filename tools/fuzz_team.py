#!/usr/bin/env python3
"""Randomised parity sweep of the four-lanes-per-instance kernels (team4 / team4v and their rollout) on the GPU box:
random members of the config-3 family - robot, scalar / matrix gains, one- or two-sided limits scaled at random,
the third task on a random subset of the joints with constant / time-dependent / input-dependent targets, feed-forward
on / off, damping 1e-9 .. 1e-5, inputs interior / mixed / near-singular - against the numpy oracle.
    python tools/fuzz_team.py [n_skills] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                   # noqa: E402

import casclik_amd as cc                             # noqa: E402
from casclik_amd import skills, sym as cs            # noqa: E402
from oracle import clik_oracle                       # noqa: E402

FK = {"iiwa": skills.iiwa(), "ur5": skills.ur5()}


def draw(rng):
    """one random member of the config-3 family with its inputs (consumes the generator exactly as the sweep does: the
    regression test of the sweep's worst case replays the stream up to its skill)"""
    robot = "ur5" if rng.random() < 0.4 else "iiwa"
    fk = FK[robot]
    n = len(fk["joint_names"])
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7 + n)
    T = fk["T_fk"](q)
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    scale = rng.uniform(0.5, 1.0)
    kw = dict(label="limits", expression=q, priority=0, set_max=scale * hi)
    if rng.random() < 0.75:
        kw["set_min"] = scale * lo
    limits = cc.SetConstraint(**kw)
    m = 6 if rng.random() < 0.7 else 3
    expr = skills._pose_expression(T, y) if m == 6 else T[:3, 3] - y[:3]
    if rng.random() < 0.3:
        expr = expr + cs.vertcat(*([0.02 * cs.sin(0.7 * t)] + [0.0] * (m - 1)))
    K = float(rng.uniform(1.0, 12.0)) if rng.random() < 0.6 else \
        np.diag(rng.uniform(1.0, 10.0, size=m)) + 0.3 * rng.normal(size=(m, m))
    pose = cc.EqualityConstraint("task", expr, gain=K, constraint_type="soft", priority=1)
    k3 = int(rng.integers(1, n + 1))
    js = sorted(rng.choice(n, size=k3, replace=False).tolist())
    rows = []
    for j in js:
        kind = rng.integers(0, 3)
        tgt = float(rng.uniform(0.3 * lo[j], 0.3 * hi[j]))
        rows.append(q[j] - tgt if kind == 0 else (q[j] - tgt - 0.1 * cs.sin(0.5 * t) if kind == 1 else q[j] - y[7 + j]))
    K3 = float(rng.uniform(0.2, 3.0)) if rng.random() < 0.7 or k3 == 1 else np.diag(rng.uniform(0.2, 3.0, size=k3))
    third = cc.EqualityConstraint("joints", cs.vertcat(*rows), gain=K3, constraint_type="soft", priority=2)
    spec = cc.SkillSpecification("fuzz_team", t, q, input_var=y, constraints=[third, limits, pose])
    opts = {"multidim_sets": True, "feedforward": bool(rng.random() < 0.8), "damping_factor": float(10 ** rng.uniform(-9, -5))}
    B = 256
    dist = ["interior", "mixed", "mixed"][int(rng.integers(0, 3))]
    Q, Y7 = skills.synthetic_inputs(fk, B, seed=int(rng.integers(1 << 30)), distribution=dist)
    if rng.random() < 0.15:
        Q[: B // 4] = rng.normal(0.0, 1e-4, size=(B // 4, n))          # near the stretched-out singularity
    Y = np.hstack([Y7, rng.uniform(0.3 * lo, 0.3 * hi, size=(B, n))])
    tval = float(rng.uniform(0.0, 3.0))
    return dict(robot=robot, fk=fk, n=n, spec=spec, opts=opts, Q=Q, Y=Y, tval=tval, m=m, k3=k3, dist=dist, limits=limits,
                has_min=kw.get("set_min") is not None, B=B)


def main():
    n_skills = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst = 0.0
    bad = 0
    for s in range(n_skills):
        d = draw(rng)
        robot, fk, n, spec, opts, Q, Y, tval, m, k3, dist, limits, B = (d[k] for k in (
            "robot", "fk", "n", "spec", "opts", "Q", "Y", "tval", "m", "k3", "dist", "limits", "B"))
        kw = {"set_min": True} if d["has_min"] else {}
        kappa = np.zeros(B)
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, tval, Q, Y=Y, cond_out=kappa)
        # degenerate ties: a 3-row position task leaves the last joint (rotation about the tool axis) with an exactly
        # zero velocity in the oracle and +-1e-20 in a factorisation-based evaluation; the tangent-cone test of that
        # joint's limit is then decided by rounding.  Lanes whose oracle mode flips under a 1e-12 relative perturbation
        # of q are skipped (as tools/fuzz_parity.py does).
        tie = np.zeros(B, dtype=bool)
        for eps in (1e-12, -1e-12):
            _, pm = clik_oracle.pinv_solve_batch(spec, opts, tval, Q * (1.0 + eps), Y=Y)
            tie |= pm != rmode
        if m == 3:
            # ... and lanes where that joint is outside its limits (the zero is structural there, not just close)
            lim_lo = limits.set_min if kw.get("set_min") is not None else -1e10 * np.ones(n)
            tie |= (Q[:, n - 1] > np.asarray(limits.set_max)[n - 1]) | (Q[:, n - 1] < np.asarray(lim_lo)[n - 1])
        # the stated rule (tests/tolerances.py): every instance against max(FLOOR, FACTOR u kappa), kappa = the worst
        # condition number the reference's algorithm meets on it (the projectors': ~2 sigma_max^2 / lam)
        from tolerances import rtol_from_cond, ILL_POSED
        tol_b = rtol_from_cond(kappa)
        tol = float(tol_b.max())
        for values in ("1", "0"):
            os.environ["CLIK_JIT_VALUES"] = values
            ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
            ctrl.setup_problem_functions()
            variant = ctrl.kernel_variant(B)
            if "/team4" not in variant:
                print("skill %3d %-4s NOT in the team family: %s" % (s, robot, variant))
                break
            dq, _, mode = ctrl.solve_batch(tval, Q, input_var=Y)
            agree = (mode == rmode) | tie
            cmp_ = (mode == rmode) & ~tie & (tol_b < ILL_POSED)
            rel_b = np.abs(dq - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))
            err = float(rel_b[cmp_].max()) if cmp_.any() else 0.0
            over = float((rel_b / tol_b)[cmp_].max()) if cmp_.any() else 0.0
            # one rollout tick must equal the solve
            q1, dq1, m1 = ctrl.rollout_batch([tval], Q, input_var=Y, dt=1e-3)
            rerr = float(np.abs(dq1 - dq).max() / (1.0 + np.abs(dq).max()))
            worst = max(worst, err)
            flag = ""
            lerr = 0.0
            if values == "1":
                # the same instances inside a batch beyond the team kernel's range: one lane per instance with the
                # numbers compiled in ("lanev": solo_tick), tick and one rollout tick, against the team kernel's answers
                reps = 16400 // B + 1
                Qb, Yb = np.tile(Q, (reps, 1)), np.tile(Y, (reps, 1))
                big = ctrl.kernel_variant(len(Qb))
                dqb, _, modeb = ctrl.solve_batch(tval, Qb, input_var=Yb)
                same = (modeb[:B] == mode) | tie
                cmpb = (modeb[:B] == mode) & ~tie
                lrel = np.abs(dqb[:B] - dq).max(axis=1) / (1.0 + np.abs(dq).max(axis=1))
                lerr = float(lrel[cmpb].max()) if cmpb.any() else 0.0
                lover = float((lrel / tol_b)[cmpb & (tol_b < ILL_POSED)].max()) if (cmpb & (tol_b < ILL_POSED)).any() else 0.0
                qb1, dqb1, mb1 = ctrl.rollout_batch([tval], Qb, input_var=Yb, dt=1e-3)
                if not big.endswith("/lanev") or (~same).any() or lover > 1.0 or not np.array_equal(mb1, modeb) \
                        or np.abs(dqb1 - dqb).max() > 1e-9 * (1.0 + np.abs(dqb).max()):
                    bad += 1
                    flag = "   <-- MISMATCH (lanev %s err %.1e)" % (big, lerr)
            if (~agree).any() or over > 1.0 or rerr > 1e-9 or not np.array_equal(m1, mode):
                bad += 1
                flag = "   <-- MISMATCH"
            print("skill %3d %-4s m=%d third=%d %-9s ff=%d lam=%.0e %-8s modes %s ties %d wrong %d err %.2e (%.2f x tol, tol <= %.0e) rollout %.1e%s" % (
                s, robot, m, k3, variant.split("/")[-1], opts["feedforward"], opts["damping_factor"], dist,
                np.bincount(rmode + 1).tolist(), int(tie.sum()), int((~agree).sum()), err, over, tol, rerr, flag))
    print("fuzz_team: %d skills, worst relative error %.3e, mismatching runs %d" % (n_skills, worst, bad))


if __name__ == "__main__":
    main()
