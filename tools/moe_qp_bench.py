#!/usr/bin/env python3
"""Per-tick cost of the Moe-2016 'singular' QP skill (ur5_moe2016_example2.ipynb cells 6-8: general inequality rows
on the tool position + soft tracking + joint-speed limits) at B instances, for inputs inside the walls (the notebook's
regime), near them, and outside (walls and speed limits active), with the status histogram of each regime.
    python tools/moe_qp_bench.py [B]
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
from extern_skills import moe_box_skill   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
ur5 = skills.ur5()


def time_tick(tick, n=200, reps=10):
    for _ in range(20):
        tick()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tick()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            tick()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * reps) * 1e6


for soft in (False, True):
    spec, home = moe_box_skill(ur5, soft_walls=soft)
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    for name, scale in (("inside the walls", 0.03), ("near the walls", 0.10), ("outside (walls active)", 0.35)):
        Q = home + rng.normal(scale=scale, size=(B, 6))
        Qd = torch.from_numpy(Q).cuda()
        st = ctrl.solve_batch(3.0, Qd)[3].cpu().numpy()
        tick = ctrl.bind_batch(Qd)
        us = time_tick(tick)
        line = "walls %-4s  %-24s kernel %-22s %7.2f us per tick of %d instances   status histogram %s" % (
            "soft" if soft else "hard", name, ctrl.kernel_variant(B), us, B, np.bincount(st, minlength=3))
        if (st != 0).any() and (st == 0).sum() >= 64:
            # the tick of a batch is its slowest instance, and an infeasible QP (the reference raises) is the slowest:
            # the same regime with the feasible instances only (tiled to the same batch size)
            ok = np.nonzero(st == 0)[0]
            Qf = Q[np.resize(ok, B)]
            usf = time_tick(ctrl.bind_batch(torch.from_numpy(Qf).cuda()))
            line += "   feasible instances only: %.2f us" % usf
        print(line)
