#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_count.py file.s [substring ...]

Static counts per kernel: fp64 VALU, other VALU, AGPR moves, DPP moves, SALU, LDS, global memory,
waits.  (Loops and branches make static != executed; the shape-specialised kernels are straight-line
up to the cold sin/cos path, so for them the two agree within a few per cent.)"""
import collections
import re
import sys


def count(path, filters=()):
    out = []
    name, ins = None, []
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name, ins = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            if "kernel" in name and all(f in name for f in filters):
                c = collections.Counter()
                for i, rest in ins:
                    if "dpp" in rest or i.endswith("_dpp"):
                        c["dpp"] += 1
                    elif i.startswith("v_accvgpr"):
                        c["agpr_mov"] += 1
                    elif i.startswith("v_") and "f64" in i:
                        c["valu_f64"] += 1
                    elif i.startswith("v_"):
                        c["valu_other"] += 1
                    elif i.startswith("s_waitcnt") or i.startswith("s_nop"):
                        c["wait_nop"] += 1
                    elif i.startswith("s_"):
                        c["salu"] += 1
                    elif i.startswith("ds_"):
                        c["lds"] += 1
                    elif i.startswith(("global_", "buffer_", "flat_", "scratch_")):
                        c["vmem"] += 1
                    else:
                        c["other"] += 1
                out.append((name, len(ins), dict(c)))
            name = None
            continue
        s = line.strip()
        if not line.startswith("\t") or not s or s[0] in ".;/":
            continue
        parts = s.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return out


if __name__ == "__main__":
    for name, n, c in count(sys.argv[1], sys.argv[2:]):
        print("%-100s %5d  %s" % (name[:100], n, " ".join("%s=%d" % kv for kv in sorted(c.items()))))
