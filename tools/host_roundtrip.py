#!/usr/bin/env python3
"""The C ABI takes device pointers; the Python layer also accepts numpy arrays and then copies
q / y to the device and dq / mode back.  This measures that host-inclusive tick (PCIe both ways +
launch + synchronisation) next to the device-resident tick, config 3, B instances.
    python tools/host_roundtrip.py [B]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
fk = skills.iiwa()
ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
ctrl.setup_problem_functions()
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
for _ in range(20):
    ctrl.solve_batch(0.0, Q, input_var=Y)
t0 = time.perf_counter()
n = 200
for _ in range(n):
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
host = (time.perf_counter() - t0) / n * 1e6
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
tick = ctrl.bind_batch(Qd, input_var=Yd)
for _ in range(50):
    tick()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    tick()
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / 2000 * 1e6
nbytes = B * (56 + 56 + 56 + 4)
print("B = %d: numpy in / numpy out %.1f us per tick (%.2f G instance-steps/s, %.1f GB/s over PCIe incl. staging); "
      "device-resident %.2f us per tick" % (B, host, B / host * 1e-3, nbytes / host * 1e-3, dev))
