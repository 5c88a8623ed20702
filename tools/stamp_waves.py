#!/usr/bin/env python3
"""Median lifetime of a wave of the lane-per-instance pose kernel (BASELINE config 2, one wave per 64 instances) when a CU
holds ONE of them (16384 instances on 256 CUs) and when it holds FOUR, one per SIMD (65536 instances): do waves on the
SIMDs of one CU slow each other?  s_memrealtime stamps at entry and exit (-DCLIK_BODY_STAMPS).
    python tools/stamp_waves.py [workload = pose] [samples = 200]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLIK_JIT_DEFINES"] = "-DCLIK_BODY_STAMPS"
os.environ["CLIK_LANES"] = "1"

import numpy as np          # noqa: E402
import torch                # noqa: E402
import casclik_amd as cc    # noqa: E402
from casclik_amd import skills, jit   # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "pose"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fk = skills.iiwa()
for B in (4096, 16384, 32768, 65536):
    Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
    Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
    if workload == "pose":
        ctrl = cc.PseudoInverseController(skill_spec=skills.pose_skill(fk))
    else:
        ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    lib = jit.attach_values.last_library
    tick = ctrl.bind_batch(Qd, input_var=Yd)
    for _ in range(50):
        tick()
    torch.cuda.synchronize()
    waves = (B + 63) // 64
    buf = (C.c_ulonglong * (2 * waves))()
    lib.clik_jit_read_body.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    rows = []
    for _ in range(S):
        tick()
        torch.cuda.synchronize()
        assert lib.clik_jit_read_body(buf, 2 * waves) == 0
        st = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(-1, 2) * 0.01
        life = st[:, 1] - st[:, 0]
        rows.append((st[:, 1].max() - st[:, 0].min(), np.median(life), life.max(), (st[:, 0] - st[:, 0].min()).max()))
    r = np.median(np.array(rows), axis=0)
    print("%s %s B %6d (%4d waves, %.1f per CU): body %.2f us, wave lifetime median %.2f longest %.2f, last start %.2f" % (
        workload, ctrl.kernel_variant(B), B, waves, waves / 256.0, r[0], r[1], r[2], r[3]))
