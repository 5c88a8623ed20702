#!/usr/bin/env python3
"""Registers / scratch / occupancy of the value-specialised kernels of a BASELINE skill, from the compiler's resource
remarks (no GPU needed).      python tools/kernel_resources.py [stack|pose|qp] [-DFLAG ...] [--asm=listing.s] [--obj=device.o]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C                                   # noqa: E402

from casclik_amd import jit, _capi, skills           # noqa: E402
from casclik_amd.build import parse_resource_remarks, FLAGS      # noqa: E402
from casclik_amd.lowering import lower_skill         # noqa: E402
import casclik_amd as cc                             # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "qp"
flags = [a for a in sys.argv[2:] if a.startswith(("-D", "-f", "-g"))]
for a in sys.argv[2:]:
    if a.startswith("--mllvm="):                      # e.g. --mllvm=-disable-machine-licm
        flags += ["-mllvm", a.split("=", 1)[1]]
asm_out = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--asm=")]
obj_out = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--obj=")]     # device code object (llvm-objdump -d)
for a in sys.argv[2:]:
    if a.startswith("--csrc="):                       # kernel headers from a scratch copy (experiments next to a running build)
        alt = os.path.abspath(a.split("=", 1)[1])
        FLAGS[:] = [("-I" + alt) if f.startswith("-I") and f.endswith("casclik_amd/csrc") else f for f in FLAGS]
lib = _capi.load_library()
fk = skills.iiwa()
if which == "qp":
    spec = skills.qp_skill(fk)
    d = lower_skill(spec)
    cdesc = _capi.desc_to_c(d)
    qc = cc.ReactiveQPController(skill_spec=spec)
    state_w = list(qc._robot_var_weights) + list(qc._virtual_var_weights[:d.n_x])
    copts = _capi.qp_opts_to_c(qc.weight_shifter, state_w, list(qc._slack_var_weights))
    buf = C.create_string_buffer(16384)
    assert lib.clik_qp_shape_describe(C.byref(cdesc), buf, len(buf)) == 1
    init, template = buf.value.decode(), jit._QP_VALUE_TEMPLATE
    words = jit.host_image_words(lib, "qp", cdesc, copts)
else:
    spec = skills.stack_skill(fk) if which == "stack" else skills.pose_skill(fk)
    opts = dict(skills.STACK_OPTIONS) if which == "stack" else {}
    d = lower_skill(spec)
    cdesc = _capi.desc_to_c(d)
    copts = _capi.pinv_opts_to_c(cc.PseudoInverseController(skill_spec=spec, options=opts).options)
    ok, init = jit.shape_of(lib, cdesc, copts)
    template = jit._VALUE_TEMPLATE
    words = jit.host_image_words(lib, "pinv", cdesc, copts)
# (the scheduling strategy the shipped object is compiled with, casclik_amd/jit.py::sched_strategy - unless one is given)
sched = jit.sched_strategy(template, init)
if sched and not any("sched-strategy" in f for f in flags):
    flags += jit.sched_flags(sched)
text = template.replace("%(nwords)d", str(len(words))).replace("%(words)s", ", ".join(w + "ull" for w in words)) % {
    "init": init, "extern": ""}
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "k.hip")
    open(src, "w").write(text)
    out = subprocess.run([jit._hipcc()] + FLAGS + ["-DCLIK_VALUE_KERNEL"] + flags + ["-c", src, "-o", os.path.join(tmp, "k.o")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        print(out.stdout.decode()[-3000:])
        sys.exit(1)
    if asm_out:
        # (the listing tools/isa_count.py reads)
        subprocess.run([jit._hipcc()] + [f for f in FLAGS if not f.startswith("-Rpass")] + ["-DCLIK_VALUE_KERNEL"] + flags +
                       ["-S", "--cuda-device-only", src, "-o", asm_out[0]], check=True)
    if obj_out:
        subprocess.run([jit._hipcc()] + [f for f in FLAGS if not f.startswith("-Rpass")] + ["-DCLIK_VALUE_KERNEL"] + flags +
                       ["-c", "--cuda-device-only", "--no-gpu-bundle-output", src, "-o", obj_out[0]], check=True)
    for name, r in sorted(parse_resource_remarks(out.stdout.decode()).items()):
        short = name.split("(")[0][-70:]
        print("%-72s VGPR %3d AGPR %3d SGPR %3d scratch %4d occupancy %d" % (
            short, r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("SGPRs", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0)))
