#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the team kernel (four lanes per instance) on the config-3 stack.
A separate JIT build with in-kernel s_memtime stamps (never the shipped library); read the SHARES.
    python tools/stamp_team.py [interior|mixed] [batch]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLIK_NO_AOT"] = "1"
os.environ["CLIK_JIT_STAMPS"] = "1"
os.environ["CLIK_LANES"] = "4"

import numpy as np          # noqa: E402
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills, jit   # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "mixed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
fk = skills.iiwa()
ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
ctrl.setup_problem_functions()
print("kernel:", ctrl.kernel_variant(B))
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution=dist)
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
tick = ctrl.bind_batch(Qd, input_var=Yd)
for _ in range(300):
    tick()
torch.cuda.synchronize()
lib = jit.attach.last_library
n = 8 * min(B // 64, 4096)
buf = (C.c_ulonglong * n)()
lib.clik_jit_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.clik_jit_read_stamps(buf, n) == 0
st = np.array(buf[:], dtype=np.float64).reshape(-1, 8)
order = [0, 1, 2, 3, 4, 6, 5]
names = ["prologue (loads -> LDS -> barrier)", "sin/cos split + FK + task rows", "targets, set bits, Gm = JJ'",
         "role rhs, LDL', two solves, J' product", "quad exchange + cone test", "select, LDS, stores"]
tot = st[:, 5] - st[:, 0]
print("per-block median cycles (s_memtime ticks): total %.0f" % np.median(tot))
for k, nm in enumerate(names):
    d = st[:, order[k + 1]] - st[:, order[k]]
    print("  %-42s %8.0f  %5.1f %%" % (nm, np.median(d), 100 * np.median(d) / np.median(tot)))
