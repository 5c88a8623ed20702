#!/usr/bin/env python3
"""Diagnostic: how many passes does the box-QP solver (qp_box_solve, config 4) need?  The pass cap
(options["max_iter"]) is swept and the instances still unconverged (status 1) are counted: cold start, and hot
start from the previous tick's partition.
    python tools/qp_passes.py [batch]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills   # noqa: E402

fk = skills.iiwa()
spec = skills.qp_skill(fk)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
for mi in (1, 2, 3, 4, 6, 8, 10, 12, 16, 120):
    ctrl = cc.ReactiveQPController(skill_spec=spec, options={"max_iter": mi})
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    hot = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = ctrl.solve_batch(0.0, Q, input_var=Y, hot_set=hot)[3]
    ref = ctrl.solve_batch(0.0, Q, input_var=Y, hot_set=hot)
    st2 = ref[3]
    st3 = ctrl.solve_batch(0.0, Q, input_var=Y, hot_set=hot)[3]
    print("pass cap %3d: unconverged cold %5d   hot (2nd tick) %5d   hot (3rd tick) %5d" % (
        mi, (st == 1).sum(), (st2 == 1).sum(), (st3 == 1).sum()))
