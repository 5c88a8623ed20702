"""Latency of the reference-style single-instance call ``controller.solve(t, q, ...)`` (B = 1):
host -> device copy, one launch, device -> host copy, DM wrapping.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import casclik_amd as cc
from casclik_amd import skills

fk = skills.iiwa()
Q, Y = skills.synthetic_inputs(fk, 4, seed=0)
for name, ctrl in (("pinv stack", cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))),
                   ("reactive qp", cc.ReactiveQPController(skill_spec=skills.qp_skill(fk)))):
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    for _ in range(50):
        ctrl.solve(0.0, Q[0], input_var=Y[0])
    t0 = time.perf_counter()
    n = 500
    for i in range(n):
        ctrl.solve(0.0, Q[i & 3], input_var=Y[i & 3])
    print("%-12s solve(): %.1f us per call" % (name, (time.perf_counter() - t0) / n * 1e6))
