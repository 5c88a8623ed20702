#!/usr/bin/env python3
"""Measurement matrix of SURVEY.md 8(d): ticks per launch K x instances per GPU B for the
config-3 stack (K = 1: one launch per tick from a hipGraph; K > 1: on-device rollout).
Prints one line per cell; run on a GPU box:  python tools/sweep.py > gpurun_out/sweep.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cells = []
for B in (4096, 16384, 131072, 1048576):
    for K in (1, 16, 256):
        steps = 1024 if B <= 131072 else 256
        warm = 256
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline", "0", "--batch", str(B),
               "--steps", str(steps), "--warmup", str(warm), "--ticks-per-launch", str(K)]
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip().splitlines()
        d = json.loads(out[-1])
        cells.append((B, K, d["roofline"]["kernel_us"], d["value"] / 1e9, d["roofline"]["frac"],
                      d["roofline"]["fp64_valu_frac_algorithmic"]))
        print("B=%8d K=%4d  %9.3f us/tick  %8.3f G instance-steps/s  hbm_frac %.4f  fp64_frac(alg) %.3f" % cells[-1],
              flush=True)
