#!/usr/bin/env python3
"""Diagnostic: where does a tick of the static stack kernel spend its cycles?
Builds the kernel with in-kernel s_memtime stamps (CLIK_JIT_STAMPS=1, a separate
JIT build - never the shipped library) and prints per-phase shares.  Read the
SHARES, not the total: the stamps fence the schedule.
    CLIK_NO_AOT=1 CLIK_JIT_STAMPS=1 python tools/stamp_profile.py [interior|mixed]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLIK_NO_AOT", "1")
os.environ.setdefault("CLIK_JIT_STAMPS", "1")

import numpy as np          # noqa: E402
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills, jit   # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "interior"
fk = skills.iiwa()
ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
ctrl.setup_problem_functions()
print("kernel:", ctrl.kernel_name)
B = 16384
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution=dist)
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
tick = ctrl.bind_batch(Qd, input_var=Yd)
for _ in range(200):
    tick()
torch.cuda.synchronize()
lib = jit.attach.last_library
n = 8 * (B // 64)
buf = (C.c_ulonglong * n)()
lib.clik_jit_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.clik_jit_read_stamps(buf, n) == 0
st = np.array(buf[:], dtype=np.float64).reshape(-1, 8)
names = ["prologue (loads -> LDS -> barrier)", "FK + task cache", "mode 0", "mode 1 (if any lane needs it)",
         "epilogue (LDS transpose + stores)"]
d = np.diff(st[:, :6], axis=1)
tot = (st[:, 5] - st[:, 0])
print("per-block median cycles (s_memtime ticks): total", np.median(tot))
for k, nm in enumerate(names):
    print("  %-40s %8.0f  %5.1f %%" % (nm, np.median(d[:, k]), 100 * np.median(d[:, k]) / np.median(tot)))
print("  prologue split: kernel args %.0f | global loads %.0f | LDS transpose + barrier %.0f" % (
    np.median(st[:, 6] - st[:, 0]), np.median(st[:, 7] - st[:, 6]), np.median(st[:, 1] - st[:, 7])))
