#!/usr/bin/env python3
"""Static per-phase instruction budget of a value-specialised kernel (no GPU needed).

    python tools/phase_budget.py [stack|pose|qp] [kernel-name substring] [-DFLAG ...] [--md=profiles/x.md]

Compiles the BASELINE skill's value-specialised instantiation with -g and books every instruction of the listing on a
phase of the tick: the `.loc` comment in front of it carries its inlining chain (innermost frame first), the kernel
headers carry CLIK_PHASE("name") / CLIK_PHASE_END() annotations (they expand to nothing), and an instruction belongs to
the last mark above the innermost frame that lies inside a marked function.  What is counted is the shipped object
itself (-g changes no instruction: the total is checked against a build without it).  The kernels are straight-line
up to the cold huge-argument sin / cos block (placed behind s_endpgm: `cold`, never executed on bench inputs) and the
cone test's skip, so static counts are executed counts."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(ins, rest):
    if "dpp" in rest or ins.endswith("_dpp"):
        return "dpp"
    if ins.startswith("v_accvgpr"):
        return "agpr_mov"
    if ins.startswith("v_") and "f64" in ins:
        if ins.startswith(("v_fma_f64", "v_fmac_f64")):
            return "fma"
        if ins.startswith(("v_mul_f64", "v_add_f64")):
            return "mul_add"
        return "f64_other"          # rcp, rsq, cmp, rndne, cvt, ldexp, div_*
    if ins.startswith("v_"):
        return "valu_32"
    if ins.startswith(("s_waitcnt", "s_nop")):
        return "wait_nop"
    if ins.startswith("s_"):
        return "salu"
    if ins.startswith("ds_"):
        return "lds"
    if ins.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


COLS = ["fma", "mul_add", "f64_other", "valu_32", "dpp", "agpr_mov", "salu", "vmem", "lds", "wait_nop", "other"]


CSRC = [os.path.join(ROOT, "casclik_amd", "csrc")]      # (--csrc=DIR: a scratch copy of the kernel headers)


def source_marks():
    """{header file name: [(line, phase name or None for CLIK_PHASE_END)]}"""
    marks = {}
    csrc = CSRC[0]
    for fn in os.listdir(csrc):
        if not fn.endswith((".hpp", ".hip")):
            continue
        rows = []
        for no, line in enumerate(open(os.path.join(csrc, fn), errors="replace"), 1):
            m = re.search(r'^\s*CLIK_PHASE\("([^"]+)"\);', line)
            if m:
                rows.append((no, m.group(1)))
            elif re.search(r"^\s*CLIK_PHASE_END\(\);", line):
                rows.append((no, None))
        if rows:
            marks[fn] = rows
    return marks


def phase_of(chain, marks):
    """chain: [(file, line)] innermost first -> the phase of the innermost frame inside a marked region"""
    for fn, line in chain:
        rows = marks.get(os.path.basename(fn))
        if not rows:
            continue
        last = None
        for no, name in rows:
            if no > line:
                break
            last = (no, name)
        if last is not None and last[1] is not None:
            return last[1]
    return "(unmarked)"


DUMP = []        # phases whose instructions are printed (--dump=phase[,phase])


def budget(listing, name_filter):
    """{kernel: [(phase, Counter)]} in order of first appearance; `cold`: code placed behind s_endpgm"""
    marks = source_marks()
    out = {}
    name, rows, ended, phase = None, None, False, "(unmarked)"
    for line in open(listing):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name = m.group(1)
            if "kernel" not in name or not all(f in name for f in name_filter):
                name = None
                continue
            rows, ended, phase = collections.OrderedDict(), False, "(unmarked)"
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = list(rows.items())
            name = None
            continue
        s = line.strip()
        if s.startswith(".loc"):
            chain = re.findall(r"([\w./-]+\.(?:hpp|hip|h)):(\d+):\d+", s.split(";", 1)[1]) if ";" in s else []
            phase = phase_of([(f, int(n)) for f, n in chain], marks)
            continue
        if not line.startswith("\t") or not s or s[0] in ".;/":
            continue
        parts = s.split(None, 1)
        ins, rest = parts[0], parts[1] if len(parts) > 1 else ""
        rows.setdefault("cold" if ended else phase, collections.Counter())[classify(ins, rest)] += 1
        if phase in DUMP and not ended:
            print("%-16s %s" % (phase, s))
        if ins == "s_endpgm":
            ended = True
    return out


def table(rows):
    lines = ["| phase | total | " + " | ".join(COLS) + " |", "|---|---|" + "---|" * len(COLS)]
    tot = collections.Counter()
    merged = collections.OrderedDict(rows)
    for ph, c in merged.items():
        n = sum(c.values())
        if ph != "cold":
            tot.update(c)
        lines.append("| %s | %d | " % (ph, n) + " | ".join(str(c.get(k, 0)) if c.get(k, 0) else "" for k in COLS) + " |")
    lines.append("| **executed (all but cold)** | **%d** | " % sum(tot.values()) + " | ".join(str(tot.get(k, 0)) for k in COLS) + " |")
    return "\n".join(lines), sum(tot.values())


def main():
    args = sys.argv[1:]
    which = args[0] if args and not args[0].startswith("-") else "stack"
    filt = [a for a in args[1:] if not a.startswith("-")]
    flags = [a for a in args if a.startswith(("-D", "-f", "--csrc="))]
    for a in args:
        if a.startswith("--csrc="):
            CSRC[0] = os.path.abspath(a.split("=", 1)[1])
    md = [a.split("=", 1)[1] for a in args if a.startswith("--md=")]
    for a in args:
        if a.startswith("--dump="):
            DUMP.extend(a.split("=", 1)[1].split(","))
    with tempfile.TemporaryDirectory() as tmp:
        marked, plain = os.path.join(tmp, "m.s"), os.path.join(tmp, "p.s")
        for out, extra in ((marked, ["-g"]), (plain, [])):
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), which] + flags + extra +
                           ["--asm=" + out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        b = budget(marked, filt)
        del DUMP[:]
        p = budget(plain, filt)
        text = []
        for name, rows in b.items():
            t, n = table(rows)
            shipped = sum(sum(c.values()) for ph, c in p.get(name, []) if ph != "cold")
            short = re.sub(r"^_ZN4clik\d+", "", name)[:60]
            text.append("### `%s`\n\n%s\n\nthe same kernel compiled without -g: %d instructions outside the cold block\n" % (short, t, shipped))
        text = "\n".join(text)
        print(text)
        if md:
            with open(md[0], "w") as f:
                f.write(text)


if __name__ == "__main__":
    main()
