#!/usr/bin/env python3
"""The four-waves-per-64-instances QP kernel (CLIK_QP_FOLIO=1, clik_qp_static.hpp: FOLIO) on BASELINE config 4: the same
answer on every run (which wave's result is taken does not depend on timing), the oracle's answer under the stated rule,
statuses equal, and how far it is from the lone-wave kernel's answer (another start of the same active-set passes).
    python tools/qp_folio_check.py [B = 16384]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np
    import casclik_amd as cc
    from casclik_amd import skills
    B = int(sys.argv[1])
    fk = skills.iiwa()
    Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
    ctrl = cc.ReactiveQPController(skill_spec=skills.qp_skill(fk))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    outs = [ctrl.solve_batch(0.0, Q, input_var=Y, use_hot=False) for _ in range(3)]
    same = all(np.array_equal(outs[0][k], o[k], equal_nan=True) for o in outs[1:] for k in (0, 2, 3))
    np.savez(sys.argv[3], dq=outs[0][0], slack=outs[0][2], status=outs[0][3], same=same)
    print(ctrl.kernel_variant(B), "three runs identical:", same)
    sys.exit(0)

import numpy as np                                   # noqa: E402
if "--delays" in sys.argv:
    # the same answers when one of the four waves starts 25 us late (-DCLIK_QP_FOLIO_DELAY=<wave>): four more builds
    B = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if len(sys.argv) > 2 else 4096
    outs = {}
    for tag, defines in [("none", "")] + [("wave %d late" % w, "-DCLIK_QP_FOLIO_DELAY=%d" % w) for w in range(4)]:
        path = "/tmp/qp_folio_delay_%s.npz" % tag.replace(" ", "_")
        subprocess.run([sys.executable, os.path.abspath(__file__), str(B), "child", path],
                       env=dict(os.environ, CLIK_QP_FOLIO="1", CLIK_JIT_DEFINES=defines), check=True)
        outs[tag] = np.load(path)
    ref = outs["none"]
    for tag, o in outs.items():
        same = all(np.array_equal(ref[k], o[k], equal_nan=True) for k in ("dq", "slack", "status"))
        print("%-12s answers bit-equal to the undelayed build: %s" % (tag, same))
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
res = {}
for folio in ("1", "0"):
    path = "/tmp/qp_folio_%s.npz" % folio
    subprocess.run([sys.executable, os.path.abspath(__file__), str(B), "child", path], env=dict(os.environ, CLIK_QP_FOLIO=folio),
                   check=True)
    res[folio] = np.load(path)
a, b = res["1"], res["0"]
print("statuses equal to the lone-wave kernel's:", np.array_equal(a["status"], b["status"]), np.bincount(a["status"], minlength=3))
ok = a["status"] == 0
rel = np.abs(a["dq"][ok] - b["dq"][ok]).max(axis=1) / (1 + np.abs(b["dq"][ok]).max(axis=1))
print("against the lone-wave kernel: worst relative difference %.2e, bit-equal instances %d of %d" % (
    rel.max(), int((rel == 0).sum()), int(ok.sum())))
from casclik_amd import skills                       # noqa: E402
from oracle import clik_oracle                       # noqa: E402
from tolerances import worst_over_tol                # noqa: E402
n = min(B, 2048)
fk = skills.iiwa()
Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
rdq, _, rsl, rst = clik_oracle.qp_solve_batch(skills.qp_skill(fk), 0.0, Q[:n], Y=Y[:n])
print("against the oracle (first %d): statuses equal %s, worst err / tol %.3f (err %.2e)" % (
    (n, np.array_equal(rst, a["status"][:n])) + worst_over_tol(np.where((rst == 0)[:, None], a["dq"][:n], 0.0), rdq, rows=rst == 0)[:2]))
print("deterministic:", bool(a["same"]))
