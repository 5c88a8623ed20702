#!/usr/bin/env python3
"""ONE resident launch of BASELINE config 3 (16384 instances, ring of four slots, every ticket valid before the kernel
starts) - the command behind the PMC passes of the resident kernel (a counter run serialises kernels: no producer may
have to run next to it):   rocprofv3 --pmc ... -- python3 tools/resident_once.py [ticks=20000] [state | pose | qp]
(state: the state kept by the kernel; pose: config 2's resident kernel; qp: config 4's, 16320 instances)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills   # noqa: E402

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
WHAT = sys.argv[2] if len(sys.argv) > 2 else "stack"
STATE = WHAT == "state"
B, RING = (16384 - 64 if WHAT == "qp" else 16384), 4
fk = skills.iiwa()
if WHAT == "qp":
    ctrl = cc.ReactiveQPController(skill_spec=skills.qp_skill(fk))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
elif WHAT == "pose":
    ctrl = cc.PseudoInverseController(skill_spec=skills.pose_skill(fk))
    ctrl.setup_problem_functions()
else:
    ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
slots = [skills.synthetic_inputs(fk, B, seed=17 * s, distribution="mixed") for s in range(RING)]
Qr = torch.stack([torch.from_numpy(q).cuda() for q, _ in slots]).contiguous()
Yr = torch.stack([torch.from_numpy(y).cuda() for _, y in slots]).contiguous()
torch.cuda.synchronize()
run = ctrl.resident_start(Qr, Yr, NT, timeout_s=3.0, ring_depth=RING, publish_ahead=NT,
                          **(dict(integrate_dt=1e-3, max_speed=2.0) if STATE else {}))
run["stream"].synchronize()
tk = run["ticket"].cpu()
print("ticks done %d of %d, stop %d" % (int(tk[49]), NT, int(tk[32])))
