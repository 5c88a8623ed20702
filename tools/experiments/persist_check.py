#!/usr/bin/env python3
"""The persistent form of the value-specialised lane kernel (clik_pinv_kernels.hpp, CLIK_LANE_PERSIST_MIN) must return
bit-for-bit what the one-chunk-per-wave launch returns: same arithmetic, another walk over the batch.
    python tools/persist_check.py [B = 1048576 + 37] -> prints a digest; run once per setting and compare, or
    python tools/persist_check.py --both [B]         -> runs itself under both settings and compares the digests"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if "--both" in sys.argv:
    args = [a for a in sys.argv[1:] if a != "--both"]
    outs = []
    for setting in ("1", str(1 << 60)):
        env = dict(os.environ, CLIK_LANE_PERSIST_MIN=setting)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, env=env, stdout=subprocess.PIPE, check=True)
        outs.append(r.stdout.decode().strip().splitlines()[-1])
        print("CLIK_LANE_PERSIST_MIN=%s: %s" % (setting, outs[-1]))
    print("IDENTICAL" if outs[0] == outs[1] else "DIFFERENT")
    sys.exit(0 if outs[0] == outs[1] else 1)

import numpy as np                                   # noqa: E402
import casclik_amd as cc                             # noqa: E402
from casclik_amd import skills                       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 20) + 37
which = sys.argv[2] if len(sys.argv) > 2 else "stack"
fk = skills.iiwa()
spec = skills.stack_skill(fk) if which == "stack" else skills.pose_skill(fk)
ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS) if which == "stack" else None)
ctrl.setup_problem_functions()
Q0, Y0 = skills.synthetic_inputs(fk, 8192, seed=3, distribution="mixed")
rng = np.random.default_rng(5)
pick = rng.integers(0, 8192, size=B)                 # (rows in a random order: every chunk differs from its neighbours)
Q, Y = Q0[pick] + 1e-3 * rng.standard_normal((B, Q0.shape[1])), Y0[pick]
dq, _, mode = ctrl.solve_batch(0.3, Q, input_var=Y)
h = hashlib.sha256(np.ascontiguousarray(dq).tobytes() + np.ascontiguousarray(mode).tobytes()).hexdigest()[:24]
print("%s B %d %s modes %s digest %s" % (which, B, ctrl.kernel_variant(B), np.bincount(mode).tolist()[:6], h))
