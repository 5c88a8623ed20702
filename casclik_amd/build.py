"""Build the HIP hot-path library in-tree (casclik_amd/libclik_hip.so).

hipcc cross-compiles for gfx950 without a GPU; the objects are cached under
casclik_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libclik_hip.so")
SOURCES = ["clik_api.hip", "clik_pinv.hip", "clik_pinv_dyn.hip", "clik_qp.hip", "clik_qp_shapes.hip",
           "clik_qp_dyn_a.hip", "clik_qp_dyn_b.hip", "clik_qp_dyn_c.hip", "clik_qp_dyn_d.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("clik_device.hpp", "clik_pinv_static.hpp", "clik_pinv_kernels.hpp", "clik_pinv_team.hpp",
                                          "clik_qp_static.hpp", "clik_qp_resident.hpp", "clik_shapes_gen.hpp", "clik_workspace.hpp",
                                          "clik_qp_dyn.hpp")] \
    + [os.path.join(ROOT, "include", "clik.h")]
ARCH = "gfx950"
# kernarg preload: the first 14 dwords of the kernel arguments (the buffer pointers and the
# batch size, which the kernels list first) arrive in SGPRs with the wave instead of through
# a scalar load from the kernarg segment - one memory round trip less at kernel start
# DEVICE_FP: device code only (-Xarch_device; host code keeps IEEE comparisons): a product with a structural zero of the
# skill (an axis component, an identity rotation of the chain, a unit row) may be dropped and "x + 0" is x.  IEEE
# semantics keep "0 * x" alive for the sake of x = NaN / inf and of the sign of a zero - in the value-specialised
# kernels that was 100 - 140 of 870 - 2500 instructions per wave doing nothing (`v_fmac_f64 v, 0, w`; round 6,
# profiles/r6_zero_products.md).  No reassociation, no reciprocal maths, contraction unchanged: results for finite
# inputs are the same bits up to the sign of a zero.  CLIK_FP_STRICT=1 builds without (A/B runs).
DEVICE_FP = [] if os.environ.get("CLIK_FP_STRICT", "0") == "1" else [
    "-Xarch_device", "-fno-signed-zeros", "-Xarch_device", "-fno-honor-nans"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         "-mllvm", "-amdgpu-kernarg-preload-count=14"] + DEVICE_FP + [
         "-Rpass-analysis=kernel-resource-usage",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


RESOURCES = os.path.join(OBJ, "kernel_resources.json")


def parse_resource_remarks(text):
    """{kernel symbol: {"VGPRs", "AGPRs", "ScratchSize", "Occupancy", ...}} from the
    -Rpass-analysis=kernel-resource-usage remarks of one hipcc run."""
    res, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+?)( \[[^\]]*\])?:\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(3)
        if key == "Function Name":
            cur = res.setdefault(val, {})
        elif cur is not None:
            try:
                cur[key.replace(" ", "")] = int(val)
            except ValueError:
                cur[key.replace(" ", "")] = val
    return res


def kernel_resources():
    """The merged report written by the last build (tests assert on it: no scratch in the
    shape-specialised kernels)."""
    with open(RESOURCES) as f:
        return json.load(f)


def build_hip(force=False, verbose=False, extra_flags=()):
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in HEADERS)
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        stale = (force or not os.path.exists(op)
                 or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time))
        if stale:
            cmd = [_hipcc()] + FLAGS + list(extra_flags) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE,
                                               stderr=subprocess.STDOUT)))
    for src, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode("utf-8", "replace")))
        # per-kernel register / scratch report of this translation unit (compiler remarks)
        with open(os.path.join(OBJ, src.replace(".hip", ".resources.json")), "w") as f:
            json.dump(parse_resource_remarks(out.decode("utf-8", "replace")), f, indent=1, sort_keys=True)
    if jobs or not os.path.exists(RESOURCES):
        merged = {}
        for src in SOURCES:
            rp = os.path.join(OBJ, src.replace(".hip", ".resources.json"))
            if os.path.exists(rp):
                with open(rp) as f:
                    merged.update(json.load(f))
        with open(RESOURCES, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)
    if jobs or not os.path.exists(LIB) or force:
        # -no-hip-rt: leave the HIP runtime symbols undefined so the library
        # binds to the ONE libamdhip64 the host process already uses (torch
        # wheels bundle their own; a second runtime cannot open the device)
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-no-hip-rt"] + objs + ["-o", LIB]
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
