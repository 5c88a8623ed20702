#!/usr/bin/env python3
"""Per-tick cost of the reference notebooks' own skills on the GPU, next to the per-solve times the
notebooks print for the CasADi controllers (one instance per call, unknown CPU):
  double_pendulum_2D_comparison_of_controllers.ipynb cell 15     QP   163 us per solve
  ur5_dual_quaternion_vs_transformation_matrix.ipynb cell 37/38  Q_dist2: pinv 0.05 ms, QP 0.24 ms (average)
These skills run as generated constraint code (casclik_amd/codegen.py) inside the run-time instantiated
kernels.  Prints us per tick for B instances and the latency of the single-instance solve().
    python tools/notebook_bench.py [B]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np          # noqa: E402
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills   # noqa: E402
from extern_skills import double_pendulum_skill, dual_quaternion_skill   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
home = np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0])
ur5 = skills.ur5()
cases = [
    ("double pendulum point skill, QP (notebook: 163 us)", cc.ReactiveQPController, double_pendulum_skill(False),
     np.stack([rng.uniform(0.25 * np.pi, 0.75 * np.pi, B), rng.uniform(0.25 * np.pi, 0.75 * np.pi, B)], axis=1),
     dict(robot_var_weights=[1.0, 1.0])),
    ("UR5 dual-quaternion Q_dist2, QP (notebook: 240 us)", cc.ReactiveQPController, dual_quaternion_skill(ur5, "Q_dist2"),
     home + rng.uniform(-1, 1, size=(B, 6)), {}),
    ("UR5 dual-quaternion Q_dist2, pinv, 64 modes (notebook: 50 us)", cc.PseudoInverseController,
     dual_quaternion_skill(ur5, "Q_dist2", for_pinv=True), home + rng.uniform(-1, 1, size=(B, 6)), {}),
]
for name, klass, spec, Q, kw in cases:
    ctrl = klass(skill_spec=spec, **kw)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    Qd = torch.from_numpy(Q).cuda()
    tick = ctrl.bind_batch(Qd)
    for _ in range(50):
        tick()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()                      # (replay: the Python call overhead is not the kernel's)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tick()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(100):
            tick()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 1000 * 1e6
    for _ in range(20):
        ctrl.solve(0.0, Q[0])
    t0 = time.perf_counter()
    for _ in range(200):
        ctrl.solve(0.0, Q[0])
    one = (time.perf_counter() - t0) / 200 * 1e6
    print("%-66s kernel %-20s %8.2f us per tick of %d instances (%.2f G instance-steps/s); solve() of one instance %.1f us"
          % (name, ctrl.kernel_name, us, B, B / us * 1e-3, one))
