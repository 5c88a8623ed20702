#!/usr/bin/env python3
"""Where does the team kernel's tick go?  Times JIT builds of the kernel with one phase removed at a
time (-DCLIK_TEAM_ABLATE bits, wrong results by design) on the config-3 stack: the difference to the full
kernel is that phase's share of the tick, stalls included (in-kernel stamps cannot give this: the compiler
moves arithmetic across them).
    python tools/ablate_team.py [batch]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import casclik_amd as cc
from casclik_amd import skills
B = int(sys.argv[1])
fk = skills.iiwa()
ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
ctrl.setup_problem_functions()
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
dQ = torch.empty_like(Qd)
tick = ctrl.bind_batch(Qd, input_var=Yd, out=dQ)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(50): tick()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(1000): tick()
    for _ in range(60): g.replay()
    s.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(40): g.replay()
    b.record(s)
    s.synchronize()
print("RESULT %%s %%.3f" %% (ctrl.kernel_variant(B), a.elapsed_time(b) * 1e3 / 40000))
""" % ROOT

B = sys.argv[1] if len(sys.argv) > 1 else "16384"
cases = [("full kernel", 0), ("no cone test", 1), ("no second solve", 2), ("no LDL' / solves", 4),
         ("no Gram build", 8), ("no FK / task rows", 16), ("no back end at all (1|4|8)", 13),
         ("prologue + epilogue only (all)", 31)]
base = None
for name, bits in cases:
    env = dict(os.environ, CLIK_NO_AOT="1", CLIK_LANES="4", CLIK_JIT_DEFINES="-DCLIK_TEAM_ABLATE=%d" % bits)
    out = subprocess.run([sys.executable, "-c", CHILD, B], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("RESULT")]
    if not line:
        print(name, "FAILED", out.stderr.decode()[-400:])
        continue
    us = float(line[0].split()[2])
    base = us if base is None else base
    print("%-34s %6.3f us/tick   (%+.3f vs full)   %s" % (name, us, us - base, line[0].split()[1]))
