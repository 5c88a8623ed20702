#!/usr/bin/env python3
"""Write the skill-image words of the BASELINE skills (what jit.attach_values / attach_qp_values compile into the
value-specialised kernels) to tests/golden/*_image_words.txt (GPU box: the words come from a device handle).  With them
__graft_entry__.build() / casclik_amd.jit.prebuild_value_kernels() instantiate those kernels ahead of time, on a
machine without a GPU, under the very cache names the controllers look up at set-up (so a GPU box without hipcc still
runs them), and tests/test_build.py checks their register use.
    python tools/dump_image_words.py [output directory = tests/golden]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import casclik_amd as cc            # noqa: E402
from casclik_amd import skills      # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
fk = skills.iiwa()
for name, ctrl, fn in (
        ("stack_iiwa", cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS)), "clik_pinv_image_words"),
        ("pose_iiwa", cc.PseudoInverseController(skill_spec=skills.pose_skill(fk)), "clik_pinv_image_words"),
        ("qp_iiwa", cc.ReactiveQPController(skill_spec=skills.qp_skill(fk)), "clik_qp_image_words")):
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    buf = (C.c_uint64 * 16384)()
    n = getattr(ctrl._lib, fn)(ctrl._handle, buf, len(buf))
    assert n > 0, name
    path = os.path.join(out, "%s_image_words.txt" % name)
    with open(path, "w") as f:
        f.write(" ".join("0x%x" % int(buf[i]) for i in range(n)))
    print("%s: wrote %d words to %s (value kernel %s)" % (name, n, path, getattr(ctrl, "value_kernel", None)))
