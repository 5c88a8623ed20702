#!/usr/bin/env python3
"""Where the four-waves-per-64-instances QP kernel (FOLIO) spends its tick: s_memrealtime stamps of every wave at entry
and after its last store (-DCLIK_BODY_STAMPS, tools/stamp_body.py's method) for whatever CLIK_QP_FOLIO /
CLIK_QP_FOLIO_SAME say - run once per setting.     python tools/stamp_folio.py [B = 16384] [samples = 300]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLIK_JIT_DEFINES"] = "-DCLIK_BODY_STAMPS"

import numpy as np          # noqa: E402
import torch                # noqa: E402
import casclik_amd as cc    # noqa: E402
from casclik_amd import skills, jit   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
S = int(sys.argv[2]) if len(sys.argv) > 2 else 300
G = 64
fk = skills.iiwa()
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
ctrl = cc.ReactiveQPController(skill_spec=skills.qp_skill(fk))
ctrl.setup_problem_functions()
ctrl.setup_solver()
lib = jit.attach_qp_values.last_library
tick = ctrl.bind_batch(Qd, input_var=Yd)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    tick()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    for _ in range(G):
        tick()
for _ in range(30):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
tick_us = e0.elapsed_time(e1) * 1e3 / (G * 20)
folio = os.environ.get("CLIK_QP_FOLIO", "") != "0"
waves = ((B + 63) // 64) * (4 if folio else 1)
buf = (C.c_ulonglong * (2 * waves))()
lib.clik_jit_read_body.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
rows = []
for _ in range(S):
    g.replay()
    torch.cuda.synchronize()
    assert lib.clik_jit_read_body(buf, 2 * waves) == 0
    st = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(-1, 2) * 0.01
    t0 = st[:, 0].min()
    life = st[:, 1] - st[:, 0]
    rows.append((st[:, 1].max() - t0, np.median(st[:, 0] - t0), (st[:, 0] - t0).max(), np.median(life), life.max(), life.min()))
r = np.median(np.array(rows), axis=0)
print("FOLIO=%s SAME=%s B %d waves %d: tick %.2f us | body %.2f | wave start after the first: median %.2f, last %.2f | wave "
      "lifetime median %.2f, longest %.2f, shortest %.2f" % (os.environ.get("CLIK_QP_FOLIO", "-"), os.environ.get("CLIK_QP_FOLIO_SAME", "-"),
                                                            B, waves, tick_us, r[0], r[1], r[2], r[3], r[4], r[5]))
