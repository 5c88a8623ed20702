#!/usr/bin/env python3
"""Diagnostic: does replaying the tick launches from a hipGraph shorten the launch gap?
Captures K launches of the bound tick on a side stream (the C ABI only enqueues a kernel on the
stream it is given, so it is capturable) and compares replay with plain stream launches.
    python tools/graph_probe.py [pinv|qp] [B]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "pinv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
fk = skills.iiwa()
if which == "pinv":
    ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
else:
    ctrl = cc.ReactiveQPController(skill_spec=skills.qp_skill(fk))
ctrl.setup_problem_functions()
ctrl.setup_solver()
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
tick = ctrl.bind_batch(Qd, input_var=Yd)
K = 200


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def plain():
    for _ in range(K):
        tick()


print("kernel", ctrl.kernel_name, "B", B)
print("stream launches : %.3f us per tick" % (timed(plain, 20) / K * 1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plain()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    plain()
print("graph replay    : %.3f us per tick" % (timed(g.replay, 20) / K * 1e6))
