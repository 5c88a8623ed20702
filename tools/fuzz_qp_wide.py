#!/usr/bin/env python3
"""Random WIDE QPs - 17 ... 42 rows, up to 32 variables, up to 24 constraints: what round 3 refused -
through the dynamic-shape kernels with their work area in global memory (clik_qp.hip, CLIK_QP_GLOBAL), statuses and
minimisers against the numpy oracle under the stated rule (tests/tolerances.py).  reactive_qp.py:191-246 bounds neither.
    python tools/fuzz_qp_wide.py [n_skills = 40] [seed = 0] [instances = 96]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CLIK_FORCE_DYNAMIC"] = "1"

import numpy as np                                   # noqa: E402

import casclik_amd as cc                             # noqa: E402
from casclik_amd import skills, sym as cs            # noqa: E402
from oracle import clik_oracle                       # noqa: E402
from tolerances import worst_over_tol                # noqa: E402

n_skills = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 96
rng = np.random.default_rng(seed)
FK = {"iiwa": skills.iiwa(), "ur5": skills.ur5()}
ran = refused = mismatching = 0
worst_all = 0.0
for s in range(n_skills):
    robot = "ur5" if rng.random() < 0.5 else "iiwa"
    fk = FK[robot]
    n = len(fk["joint_names"])
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    lo, hi, vmax = np.array(fk["lower"]), np.array(fk["upper"]), np.array(fk["velocity"])
    cons, what = [], []
    pri = iter(range(1, 200))
    for axis, name in enumerate("xyz"):
        if rng.random() < 0.7:
            kind = "hard" if rng.random() < 0.6 else "soft"
            a, b = (-0.9, 0.9) if axis < 2 else (0.05, 1.4)
            cons.append(cc.SetConstraint(label="wall_" + name, expression=T[axis, 3], set_min=a, set_max=b, priority=next(pri),
                                         constraint_type=kind, gain=float(rng.uniform(2.0, 8.0))))
            what.append("wall %s %s" % (name, kind))
    if rng.random() < 0.6:
        cons.append(cc.EqualityConstraint(label="pose", expression=skills._pose_expression(T, y), gain=float(rng.uniform(1.0, 6.0)),
                                          constraint_type="soft", priority=next(pri)))
        what.append("pose")
    else:
        cons.append(cc.EqualityConstraint(label="position", expression=T[:3, 3] - y[:3], gain=float(rng.uniform(1.0, 6.0)),
                                          constraint_type="soft", priority=next(pri)))
        what.append("position")
    kind = "hard" if rng.random() < 0.7 else "soft"
    cons.append(cc.SetConstraint(label="limits", expression=q, set_min=lo, set_max=hi, priority=0, constraint_type=kind,
                                 gain=float(rng.uniform(1.0, 4.0))))
    what.append("limits " + kind)
    if rng.random() < 0.8:
        cons.append(cc.VelocitySetConstraint(label="speed", expression=q, set_min=-vmax, set_max=vmax, priority=0))
        what.append("speed")
    if rng.random() < 0.7:
        cons.append(cc.EqualityConstraint(label="posture", expression=q - float(rng.uniform(-0.3, 0.3)), gain=float(rng.uniform(0.2, 1.0)),
                                          constraint_type="soft", priority=next(pri)))
        what.append("posture")
    for j in range(int(rng.integers(0, n))):
        if len(cons) >= 22:
            break
        if rng.random() < 0.5:
            cons.append(cc.EqualityConstraint(label="rest_q%d" % j, expression=q[j] - 0.1 * (j + 1), gain=float(rng.uniform(0.2, 1.0)),
                                              constraint_type="soft", priority=next(pri)))
        else:
            cons.append(cc.VelocityEqualityConstraint(label="drift_q%d" % j, expression=q[j], target=0.01 * (j - 3),
                                                      constraint_type="soft", priority=next(pri)))
        what.append(cons[-1].label)
    spec = cc.SkillSpecification("wide", t, q, input_var=y, constraints=cons)
    Q, Y = skills.synthetic_inputs(fk, B, seed=int(rng.integers(1 << 30)), distribution="mixed" if rng.random() < 0.5 else "interior")
    tval = float(rng.uniform(0.0, 3.0))
    try:
        ctrl = cc.ReactiveQPController(skill_spec=spec)
        ctrl.setup_problem_functions()
        ctrl.setup_solver()
    except NotImplementedError as e:
        refused += 1
        print("%2d %-4s REFUSED (%s)  [%s]" % (s, robot, str(e)[:80], "; ".join(what)))
        continue
    rows, nvars = ctrl.n_qp_rows, ctrl.n_qp_vars
    if rows < 17:
        continue                                      # (not wide: the other sweeps' range)
    try:
        dq, _, slack, status = ctrl.solve_batch(tval, Q, input_var=Y)
    except Exception as e:                            # a refusal at solve time counts as one
        refused += 1
        print("%2d %-4s REFUSED at solve (%s)  rows %d vars %d [%s]" % (s, robot, str(e)[:80], rows, nvars, "; ".join(what)))
        continue
    rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(spec, tval, Q, Y=Y)
    ran += 1
    same = np.array_equal(status, rstatus)
    ok = (status == 0) & (rstatus == 0)
    ratio, err, left = worst_over_tol(np.where(ok[:, None], dq, 0.0), rdq, rows=ok) if ok.any() else (0.0, 0.0, 0)
    worst_all = max(worst_all, ratio)
    bad = (not same) or ratio > 1.0
    mismatching += int(bad)
    print("%2d %-4s %-8s rows %2d vars %2d constraints %2d  status %s (oracle %s)  err %.1e (%.2f x tol, %d ill-posed)%s  [%s]" % (
        s, robot, ctrl.kernel_name, rows, nvars, len(cons), np.bincount(status, minlength=3).tolist(),
        np.bincount(rstatus, minlength=3).tolist(), err, ratio, left, "   <-- MISMATCH" if bad else "", "; ".join(what)))
print("wide QPs: %d ran (17+ rows), %d refused, %d mismatching; worst err / tol %.3f" % (ran, refused, mismatching, worst_all))
