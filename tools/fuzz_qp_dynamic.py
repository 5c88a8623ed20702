#!/usr/bin/env python3
"""Hunt for false "infeasible" answers of the dynamic-shape QP kernel (clik_qp.hip: all soft equalities as
active-set rows).  Random skills from tools/fuzz_parity.py's pool, constraints made soft at random, through
CLIK_FORCE_DYNAMIC=1 only (no run-time instantiation: fast), statuses against the numpy oracle.  Prints one line per
skill with a status disagreement and, at the end, the (seed, skill index) pairs to turn into regression tests.
    python tools/fuzz_qp_dynamic.py [n_skills] [seed] [instances]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["CLIK_FORCE_DYNAMIC"] = "1"

import numpy as np                                   # noqa: E402

import casclik_amd as cc                             # noqa: E402
from casclik_amd import skills                       # noqa: E402
from oracle import clik_oracle                       # noqa: E402
import fuzz_parity
from tolerances import worst_over_tol                                   # noqa: E402

n_skills = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 192
rng = np.random.default_rng(seed)
FK = {"iiwa": skills.iiwa(), "ur5": skills.ur5()}
ran = bad_total = inst = 0
hits = []
for s in range(n_skills):
    robot = "ur5" if rng.random() < 0.5 else "iiwa"
    fk = FK[robot]
    n = len(fk["joint_names"])
    spec, opts, rest = fuzz_parity.random_skill(rng, fk, n)
    for c in spec.constraints:
        if isinstance(c, (cc.EqualityConstraint, cc.SetConstraint)):
            c.constraint_type = "soft" if rng.random() < 0.7 else "hard"
    spec = cc.SkillSpecification("fuzz_qp", spec.time_var, spec.robot_var,
                                 input_var=spec.input_var if spec.n_input_var > 0 else None,
                                 constraints=list(spec.constraints))
    Q, Y = skills.synthetic_inputs(fk, B, seed=int(rng.integers(1 << 30)), distribution="mixed")
    Yin = Y if spec.n_input_var > 0 else None
    tval = float(rng.uniform(0.0, 3.0))
    try:
        qc = cc.ReactiveQPController(skill_spec=spec)
        qc.setup_problem_functions()
        qc.setup_solver()
    except NotImplementedError:
        continue
    try:
        rdq, _, rsl, rst = clik_oracle.qp_solve_batch(spec, tval, Q, Y=Yin)
    except Exception:
        continue
    dq, _, sl, st = qc.solve_batch(tval, Q, input_var=Yin)
    ran += 1
    inst += B
    false_inf = (rst == 0) & (st == 2)
    missed = (rst == 2) & (st == 0)
    ok = (rst == 0) & (st == 0)
    over, err, _ = worst_over_tol(dq, rdq, rows=ok)          # (the stated rule, tests/tolerances.py)
    if false_inf.any() or over > 1.0:
        bad_total += int(false_inf.sum())
        hits.append((seed, s, robot, np.nonzero(false_inf)[0][:6].tolist()))
        print("skill %3d %-4s rows %2d tasks %s: oracle feasible & device infeasible on %d of %d (oracle infeasible %d, missed %d) err %.1e" % (
            s, robot, qc.n_qp_rows, [type(c).__name__[:6] + ":" + c.label + ":" + c.constraint_type[0] for c in spec.constraints],
            int(false_inf.sum()), B, int((rst == 2).sum()), int(missed.sum()), err))
print("ran %d dynamic-QP skills, %d instances; false infeasible: %d   seeds: %s" % (ran, inst, bad_total, hits))
