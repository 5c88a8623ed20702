#!/usr/bin/env python3
"""The reference's own simulation loop (ur5_moe2016_example2.ipynb:503-549: solve -> clamp at pi / 5 -> Euler step) for
the Moe-2016 skills of BOTH controllers - the pinv skill with three 1-D wall sets (8 modes), the pinv skill with the
multidimensional set, and the QP skills with general wall rows - at B instances, as ON-DEVICE ROLLOUTS of 256 ticks per
launch against a launch per tick.  The same controllers, through the same `rollout_batch`, retrace
the notebook's stored figures in tests/test_gpu_figure_pins.py::test_on_device_rollouts_reproduce_the_moe_2016_figures.
    python tools/moe_rollout_bench.py [B=16384] [ticks per launch=256]
The "rollout" column is HOST-INCLUSIVE: these skills follow a time trajectory, and `rollout_batch` evaluates its value and
derivative for every tick of the launch on the host (Python) and copies them over before it launches - ~20 us per tick of
host work that a loop which prepares the next launch's terms beside the running one would hide.  The DEVICE time per tick
is the rollout kernel's duration / ticks: run this script under `rocprofv3 --kernel-trace --stats` (a 256-tick launch lasts a
millisecond: the profiler's stretch of ~1 us does not matter) - profiles/r6_moe_rollout_kernel_stats.csv.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np          # noqa: E402
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
import notebook_figures as cf       # noqa: E402
import figure_skills        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
TPL = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(0)
fk = cf.moe_fk()


def per_tick_launch(ctrl, Qd, t0, n=200, reps=10):
    bound = ctrl.bind_batch(Qd)

    def tick():
        bound(t0)
    for _ in range(20):
        tick()
    torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
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
    t = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


def rollout(ctrl, Q, t_start, launches=6):
    times = t_start + cf.MOE_DT * np.arange(TPL)
    q = torch.from_numpy(Q).cuda()
    q = ctrl.rollout_batch(times, q, dt=cf.MOE_DT, max_speed=figure_skills.MOE_MAX_SPEED)[0]       # (warm-up launch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(launches):
        res = ctrl.rollout_batch(times + (k + 1) * TPL * cf.MOE_DT, q, dt=cf.MOE_DT, max_speed=figure_skills.MOE_MAX_SPEED)
        q = res[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (launches * TPL) * 1e6, res


print("Moe-2016 skills (ur5_moe2016_example2.ipynb), %d instances around the notebook's start pose; rollouts of %d ticks per launch"
      % (B, TPL))
for case in cf.MOE_CASES:
    kind, sit = case.split("_")
    spec = cf.moe_skill(fk, sit)
    if kind == "pinv":
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=cf.moe_options(case))
    else:
        ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    Q = cf.MOE_HOME[None, :] + rng.normal(scale=0.03, size=(B, 6))
    us_roll, res = rollout(ctrl, Q, 0.0)
    us_tick = per_tick_launch(ctrl, torch.from_numpy(Q).cuda(), 0.0)
    extra = ""
    if kind == "pinv":
        extra = "modes at the end %s" % np.bincount(res[2].cpu().numpy().astype(int) + 1, minlength=2)[:9]
    else:
        extra = "statuses at the end %s" % np.bincount(res[-1].cpu().numpy().astype(int), minlength=3)
    print("%-14s kernel %-26s rollout (host-inclusive) %6.2f us per tick   launch per tick %6.2f us   %s"
          % (case, ctrl.kernel_variant(B) if hasattr(ctrl, "kernel_variant") else ctrl.kernel_name, us_roll, us_tick, extra))
