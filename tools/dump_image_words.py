#!/usr/bin/env python3
"""Write the skill-image words of the config-3 stack (what jit.attach_values compiles in) to a file, so that the
value-specialised kernel can be cross-compiled and its ISA inspected on a machine without a GPU.
    python tools/dump_image_words.py gpurun_out/stack_iiwa_words.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import casclik_amd as cc            # noqa: E402
from casclik_amd import skills      # noqa: E402

ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(skills.iiwa()), options=dict(skills.STACK_OPTIONS))
ctrl.setup_problem_functions()
buf = (C.c_uint64 * 16384)()
n = ctrl._lib.clik_pinv_image_words(ctrl._handle, buf, len(buf))
assert n > 0
with open(sys.argv[1], "w") as f:
    f.write(" ".join("0x%x" % int(buf[i]) for i in range(n)))
print("wrote %d words, variant %s" % (n, ctrl.kernel_variant(16384)))
