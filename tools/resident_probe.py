#!/usr/bin/env python3
"""Resident ticks (clik_pinv_resident_run): per-tick cost of the config-3 team kernel when it stays on the device and is
fed tickets by a device-side producer, against one launch per tick (hipGraph) and the on-device rollout.

  free-running   the producer publishes all tickets at once: the kernel never waits - its own per-tick cost
                 (acquire + reload of q / y + tick + stores + release + count)
  closed loop    the producer publishes ticket k only after every wave has counted tick k-1: both hand-offs
                 (N waves -> 1 producer -> N waves) are on the critical path, as for a producer that needs dq

    python tools/resident_probe.py [B=16384] [ticks=20000]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402

import casclik_amd as cc    # noqa: E402
from casclik_amd import skills   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
fk = skills.iiwa()
ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
ctrl.setup_problem_functions()
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
ref = ctrl.solve_batch(0.0, Qd, input_var=Yd)
# a ring of RING input / output slots with a different batch in each (tick k uses slot (k - 1) % RING): what a producer
# that runs ahead of the kernel needs - with ONE buffer it could only write after every wave has finished the tick
RING = int(os.environ.get("CLIK_PROBE_RING", "4"))
slots = [(Q, Y)] + [skills.synthetic_inputs(fk, B, seed=17 * s, distribution="mixed") for s in range(1, RING)]
Qr = torch.stack([torch.from_numpy(q).cuda() for q, _ in slots]).contiguous()
Yr = torch.stack([torch.from_numpy(y).cuda() for _, y in slots]).contiguous()
refs = [ctrl.solve_batch(0.0, Qr[s], input_var=Yr[s]) for s in range(RING)]
print("kernel", ctrl.kernel_variant(B))

# one launch per tick (graph replay), for the same box
tick = ctrl.bind_batch(Qd, input_var=Yd)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    tick()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    for _ in range(1000):
        tick()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("launch per tick (hipGraph of 1000):   %.3f us per tick" % ((time.perf_counter() - t0) / 10000 * 1e6))

def run_ticks(nt, closed, integrate=False):
    """best of 5: wall time [s] from the feeder's launch to the resident kernel's exit, for nt ticks"""
    best = None
    for rep in range(5):
        torch.cuda.synchronize()
        run = ctrl.resident_start(Qr, Yr, nt, timeout_s=3.0, ring_depth=RING,
                                  **(dict(integrate_dt=1e-3, max_speed=2.0) if integrate else {}))
        feeder_stream = ctrl.resident_feed_stream()       # (one that makes progress beside the resident kernel)
        time.sleep(0.02)                      # (the resident kernel is up and polling)
        t0 = time.perf_counter()
        ctrl.resident_feed(run, nt, closed_loop=closed, timeout_s=3.0, stream=feeder_stream)
        run["stream"].synchronize()
        el = time.perf_counter() - t0
        feeder_stream.synchronize()
        tk = run["ticket"].cpu().numpy()
        dn = run["done"].cpu().numpy()
        ok = (tk[32] == 0) and (tk[49] == nt) and (dn == nt).all()
        same = all(torch.equal(run["out"][s], refs[s][0]) and torch.equal(run["mode"][s], refs[s][2]) for s in range(RING))
        if integrate:
            same = True         # (the state moves: parity of this mode is tests/test_gpu_team.py's job)
        if not (ok and same):
            print("  %s rep %d: ticks done %d, stop %d, slots at the last tick %d of %d, equal to the launched tick: %s"
                  % ("closed loop" if closed else "free-running", rep, tk[49], tk[32], int((dn == nt).sum()), run["waves"], same))
        best = el if best is None else min(best, el)
    return best


print("pipeline:", os.environ.get("CLIK_JIT_DEFINES", "") or "on (default)", " ring of %d slots, a different batch in each" % RING)
for name, closed, integ in (("fed ahead (all tickets published)", False, False),
                            ("fed ahead, state integrated in the kernel (targets from the ring)", False, True),
                            ("closed loop (ticket k after every done[k-1])", True, False)):
    # two run lengths: the slope is the per-tick cost, the intercept what a run costs around its ticks
    short, long_ = run_ticks(NT, closed, integ), run_ticks(3 * NT, closed, integ)
    slope = (long_ - short) / (2 * NT) * 1e6
    print("resident, %-70s %.3f us per tick (slope between %d and %d ticks; the runs as wholes: %.3f / %.3f us per tick, "
          "fixed part %.0f us)" % (name, slope, NT, 3 * NT, short / NT * 1e6, long_ / (3 * NT) * 1e6,
                                  (short - slope * 1e-6 * NT) * 1e6))
