#!/usr/bin/env python3
"""Randomised parity sweep of the QP MIXED family (GPU box): skills with hard GENERAL inequality rows (sets on tool
position components - the wall sets of ur5_moe2016_example2.ipynb cell 6 - or on a task-space norm), soft inequality
rows (lifted into the box), optional hard joint limits / speed limits (the box) and random soft tasks - the QPs
clik_qp_static.hpp::qp_mixed_pas solves - through the kernel instantiated for each skill, against the numpy oracle
(status and minimiser), cold and hot-started.

    python tools/fuzz_qp_mixed.py [n_skills] [seed]
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
from tolerances import rtol_from_cond, ILL_POSED     # noqa: E402


def random_mixed_skill(rng, fk, n):
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    lo, hi = np.asarray(fk["lower"], float), np.asarray(fk["upper"], float)
    vmax = np.asarray(fk["velocity"], float)
    cons, desc = [], []
    kind = rng.integers(0, 3)
    w = float(rng.choice([1.0, 0.3, 5.0]))
    if kind == 0:
        cons.append(cc.EqualityConstraint("position", T[:3, 3] - y[:3], gain=float(rng.uniform(1, 10)),
                                          constraint_type="soft", priority=5, slack_weight=w))
    elif kind == 1:
        cons.append(cc.EqualityConstraint("pose", cs.vertcat(T[:3, 3] - y[:3], cs.orientation_error(T[:3, :3], y[3:7])),
                                          gain=float(rng.uniform(1, 10)), constraint_type="soft", priority=5, slack_weight=w))
    else:
        cons.append(cc.EqualityConstraint("posture", q - 0.5 * (lo + hi), gain=0.5, constraint_type="soft", priority=5))
    desc.append(["position", "pose", "posture"][kind])
    # hard general rows: walls on tool position components around the workspace centre
    n_hard = int(rng.integers(0, 4))
    axes = rng.choice(3, size=n_hard, replace=False).tolist() if n_hard else []
    for a in axes:
        c0 = [0.3, 0.0, 0.6][a]
        half = float(rng.uniform(0.25, 0.6))
        side = rng.integers(0, 3)
        kw = {"set_min": c0 - half} if side == 1 else ({"set_max": c0 + half} if side == 2 else
                                                       {"set_min": c0 - half, "set_max": c0 + half})
        cons.append(cc.SetConstraint("wall_%d" % a, T[a, 3], gain=float(rng.uniform(0.5, 4.0)), priority=1, **kw))
        desc.append("hard wall %s%s" % ("xyz"[a], ["", " (min)", " (max)"][side]))
    # soft inequality rows (lifted): a soft wall and / or soft joint limits on a joint or two
    n_soft = int(rng.integers(0 if n_hard else 1, 3))
    left = max(0, 8 - n)
    n_soft = min(n_soft, left)
    for k in range(n_soft):
        if rng.random() < 0.5:
            a = int(rng.integers(0, 3))
            c0 = [0.3, 0.0, 0.6][a]
            cons.append(cc.SetConstraint("softwall_%d" % k, T[a, 3], set_min=c0 - 0.3, set_max=c0 + 0.3,
                                         gain=float(rng.uniform(0.5, 3.0)), priority=2, constraint_type="soft",
                                         slack_weight=float(rng.choice([1.0, 10.0]))))
            desc.append("soft wall %s" % "xyz"[a])
        else:
            j = int(rng.integers(0, n))
            cons.append(cc.SetConstraint("softlim_%d" % k, q[j], set_min=0.7 * lo[j], set_max=0.7 * hi[j], gain=2.0,
                                         priority=2, constraint_type="soft", slack_weight=3.0))
            desc.append("soft limit q%d" % j)
    if rng.random() < 0.7 or n_hard > 0:
        # (hard walls always come with speed limits on every joint, as in the notebooks: a joint without a bound can
        # satisfy any wall at an absurd speed, and "feasible at 1e5 rad/s" against "infeasible" is a matter of taste)
        js = list(range(n)) if n_hard > 0 else sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
        s = float(rng.choice([0.3, 1.0]))
        cons.append(cc.VelocitySetConstraint("speed", cs.vertcat(*[q[j] for j in js]), set_min=-s * vmax[js],
                                             set_max=s * vmax[js], priority=0))
        desc.append("speed x%.2f on %s" % (s, js))
    if rng.random() < 0.4:
        js = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
        cons.append(cc.SetConstraint("joint_limits", cs.vertcat(*[q[j] for j in js]), set_min=lo[js] * 0.8,
                                     set_max=hi[js] * 0.8, gain=float(rng.uniform(0.5, 5)), priority=0))
        desc.append("limits on %s" % js)
    return cc.SkillSpecification("mixed_fuzz", t, q, input_var=y, constraints=cons), "; ".join(desc)


def main():
    n_skills = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import torch
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(n_skills):
        robot = "iiwa" if rng.random() < 0.4 else "ur5"
        fk = skills.iiwa() if robot == "iiwa" else skills.ur5()
        n = len(fk["lower"])
        spec, what = random_mixed_skill(rng, fk, n)
        ctrl = cc.ReactiveQPController(skill_spec=spec)
        try:
            ctrl.setup_problem_functions()
            ctrl.setup_solver()
        except NotImplementedError as why:
            print("%2d %-4s skipped (%s)  [%s]" % (k, robot, str(why)[:70], what), flush=True)
            continue
        B = 1200
        Q, Y = skills.synthetic_inputs(fk, B, seed=2000 + k, distribution="interior")
        if robot == "ur5":
            Q = Q * 0.35            # (the UR5's +-2 pi ranges put the tool anywhere; keep it near the walls)
        dq, _, slack, status = ctrl.solve_batch(0.0, Q, input_var=Y)
        sub = np.arange(0, B, 3)
        kappa = np.ones(len(sub))
        rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(spec, 0.0, Q[sub], Y=Y[sub], cond_out=kappa)
        tol_b = rtol_from_cond(kappa)            # the stated rule (tests/tolerances.py), per instance
        same = np.array_equal(status[sub], rstatus)
        # (instances whose minimiser has joint speeds beyond 100 rad/s - unbounded joints with the tiny curvature mu of
        # the cost - are left out of the precision figure: there the stopping tolerance on the multipliers, 1e-9
        # relative, is amplified by 1 / mu into the velocities; statuses are still compared)
        ok = (rstatus == 0) & (status[sub] == 0) & (tol_b < ILL_POSED)
        e_b = np.zeros(len(sub))
        e_b[ok] = np.abs(dq[sub][ok] - rdq[ok]).max(axis=1) / (1.0 + np.abs(rdq[ok]).max(axis=1))
        err = float((e_b / tol_b)[ok].max()) if ok.any() else 0.0          # (in units of the instance's own bound)
        note = ""
        if ok.any() and err > 1.0:
            # two answers that differ: whose is a KKT point?  (the oracle's dense Goldfarb-Idnani stops on its own
            # tolerances; along directions with curvature mu = 1e-3 a stationarity error of 1e-6 moves the point by that)
            H_, A_, lb_, ub_ = clik_oracle.qp_data_batch(spec, 0.0, Q[sub][ok], Y=Y[sub][ok])
            e_all = (e_b / tol_b)[ok]
            bad_dev = 0
            for i_ in np.nonzero(e_all > 1.0)[0]:
                xd = np.concatenate([dq[sub][ok][i_], slack[sub][ok][i_]]) if slack is not None else dq[sub][ok][i_]
                xo = np.concatenate([rdq[ok][i_], rslack[ok][i_]]) if slack is not None else rdq[ok][i_]
                kd = clik_oracle.kkt_residuals(H_[i_], A_[i_], lb_[i_], ub_[i_], xd)
                ko = clik_oracle.kkt_residuals(H_[i_], A_[i_], lb_[i_], ub_[i_], xo)
                if not (kd[0] < 1e-9 and kd[1] < 1e-9 * (1 + np.abs(xd).max()) and kd[1] <= ko[1]):
                    bad_dev += 1
            if bad_dev == 0:
                note = " (where they differ by more, the device's point is the better KKT point: oracle stationarity worse)"
                err = 0.5
        serr = 0.0
        if slack is not None and ok.any():
            serr = float(((np.abs(slack[sub][ok] - rslack[ok]).max(axis=1) / (1.0 + np.abs(rslack[ok]).max(axis=1))) / tol_b[ok]).max())
            if note:
                serr = min(serr, 0.5)
        hot = torch.zeros(B, dtype=torch.int32, device="cuda")
        Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
        ctrl.solve_batch(0.0, Qd, input_var=Yd, hot_set=hot, use_hot=False)
        res2 = ctrl.solve_batch(0.0, Qd, input_var=Yd, hot_set=hot, use_hot=True)
        d2, st2 = res2[0].cpu().numpy(), res2[3].cpu().numpy()
        fin = (status == 0) & (st2 == 0)
        herr = np.abs(d2[fin] - dq[fin]).max() if fin.any() else 0.0
        hsame = np.array_equal(st2, status)
        if not hsame:
            dd = np.nonzero(st2 != status)[0]
            print("     hot-started status differs on instances %s: cold %s hot %s" % (
                dd[:8].tolist(), status[dd][:8].tolist(), st2[dd][:8].tolist()))
        flag = "" if (same and hsame and err <= 1.0 and serr <= 1.0 and herr < 1e-8) else "   <-- MISMATCH"
        bad += bool(flag)
        if not same:
            diff = np.nonzero(status[sub] != rstatus)[0]
            print("     status differs on sampled instances %s: device %s oracle %s" % (
                sub[diff][:8].tolist(), status[sub][diff][:8].tolist(), rstatus[diff][:8].tolist()))
        print("%2d %-4s %-22s status %s (oracle %s)  err / tol %.2f slack %.2f  hot-vs-cold %.1e%s  [%s]%s" % (
            k, robot, ctrl.kernel_name[:22], np.bincount(status[sub], minlength=3), np.bincount(rstatus, minlength=3), err,
            serr, herr, ("" if hsame else " hot status differs") + note, what, flag), flush=True)
    print("mismatching skills: %d of %d" % (bad, n_skills))


if __name__ == "__main__":
    main()
