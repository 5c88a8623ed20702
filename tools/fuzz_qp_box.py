#!/usr/bin/env python3
"""Randomised parity sweep of the QP box family (GPU box): skills whose hard rows are all bounds on single states
(joint limits / speed limits on random subsets of the joints, one- or two-sided, random gains) behind random soft
tasks - the QPs clik_qp_static.hpp solves with Gauss-Seidel sweeps + the primal active set (qp_box_pas) - through the
kernel instantiated for each skill, against the numpy oracle (status and minimiser), cold and hot-started.

    python tools/fuzz_qp_box.py [n_skills] [seed]
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
from tolerances import worst_over_tol                # noqa: E402


def random_box_skill(rng, fk, n):
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    lo, hi = np.asarray(fk["lower"], float), np.asarray(fk["upper"], float)
    vmax = np.asarray(fk["velocity"], float)
    cons, desc = [], []
    kind = rng.integers(0, 4)
    w = float(rng.choice([1.0, 0.3, 5.0]))
    if kind == 0:
        cons.append(cc.EqualityConstraint("position", T[:3, 3] - y[:3], gain=float(rng.uniform(1, 10)),
                                          constraint_type="soft", priority=1, slack_weight=w))
    elif kind == 1:
        cons.append(cc.EqualityConstraint("pose", cs.vertcat(T[:3, 3] - y[:3], cs.orientation_error(T[:3, :3], y[3:7])),
                                          gain=float(rng.uniform(1, 10)), constraint_type="soft", priority=1, slack_weight=w))
    elif kind == 2:
        cons.append(cc.EqualityConstraint("position", T[:3, 3] - y[:3], gain=3.0, constraint_type="soft", priority=1))
        cons.append(cc.EqualityConstraint("posture", q - 0.5 * (lo + hi), gain=0.5, constraint_type="soft", priority=2,
                                          slack_weight=0.1))
    else:
        cons.append(cc.EqualityConstraint("height", T[2, 3] - y[2], gain=4.0, constraint_type="soft", priority=1))
    desc.append(["position", "pose", "position+posture", "height"][kind])
    if rng.random() < 0.75:
        js = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
        sel = cs.vertcat(*[q[j] for j in js])
        side = rng.integers(0, 3)
        kw = {"set_min": lo[js] * 0.8} if side == 1 else ({"set_max": hi[js] * 0.8} if side == 2 else
                                                         {"set_min": lo[js] * 0.8, "set_max": hi[js] * 0.8})
        cons.append(cc.SetConstraint("joint_limits", sel, gain=float(rng.uniform(0.5, 20)), priority=0, **kw))
        desc.append("limits%s on %s" % (["", " (lower only)", " (upper only)"][side], js))
    if rng.random() < 0.8 or len(cons) == 1:
        js = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
        s = float(rng.choice([0.05, 0.3, 1.0]))
        cons.append(cc.VelocitySetConstraint("speed", cs.vertcat(*[q[j] for j in js]), set_min=-s * vmax[js],
                                             set_max=s * vmax[js], priority=0))
        desc.append("speed x%.2f on %s" % (s, js))
    return cc.SkillSpecification("box_fuzz", t, q, input_var=y, constraints=cons), "; ".join(desc)


def main():
    n_skills = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import torch
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(n_skills):
        robot = "iiwa" if rng.random() < 0.5 else "ur5"
        fk = skills.iiwa() if robot == "iiwa" else skills.ur5()
        n = len(fk["lower"])
        spec, what = random_box_skill(rng, fk, n)
        ctrl = cc.ReactiveQPController(skill_spec=spec)
        ctrl.setup_problem_functions()
        ctrl.setup_solver()
        B = 1500
        Q, Y = skills.synthetic_inputs(fk, B, seed=1000 + k, distribution="mixed")
        dq, _, slack, status = ctrl.solve_batch(0.0, Q, input_var=Y)
        sub = np.arange(0, B, 3)
        rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(spec, 0.0, Q[sub], Y=Y[sub])
        same = np.array_equal(status[sub], rstatus)
        ok = rstatus == 0
        # (the stated rule, tests/tolerances.py: every instance against max(FLOOR, FACTOR u kappa) of ITS QP)
        over, err, _ = worst_over_tol(dq[sub], rdq, rows=ok & (status[sub] == 0))
        hot = torch.zeros(B, dtype=torch.int32, device="cuda")
        Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
        ctrl.solve_batch(0.0, Qd, input_var=Yd, hot_set=hot, use_hot=False)
        d2 = ctrl.solve_batch(0.0, Qd, input_var=Yd, hot_set=hot, use_hot=True)[0].cpu().numpy()
        fin = status == 0
        herr = np.abs(d2[fin] - dq[fin]).max() if fin.any() else 0.0
        flag = "" if (same and over <= 1.0 and herr < 1e-8) else "   <-- MISMATCH (%.2f x the stated tolerance)" % over
        bad += bool(flag)
        print("%2d %-4s %-26s rows %2d  status %s  rel err %.1e  hot-vs-cold %.1e  [%s]%s" % (
            k, robot, ctrl.kernel_name[:26], ctrl.n_rows if hasattr(ctrl, "n_rows") else -1, np.bincount(status, minlength=3),
            err, herr, what, flag), flush=True)
    print("mismatching skills: %d of %d" % (bad, n_skills))


if __name__ == "__main__":
    main()
