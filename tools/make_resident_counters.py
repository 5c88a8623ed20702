#!/usr/bin/env python3
"""profiles/r6_boundary_free_counters.json / .md from the PMC passes of tools/profile_round6.sh: the on-device rollouts and
the resident tick kernels of both controllers, PER TICK AND WAVE.  One or more result directories,
each `label=dir` (e.g. flags=gpurun_out/r6prof_flags diet=gpurun_out/r6prof_diet final=gpurun_out/r6prof_final):
    python tools/make_resident_counters.py profiles/r6_boundary_free_counters label=dir [label=dir ...]
SQ_*_CYCLES-type counters tick once per 4 clocks (MI355X_MICROARCH.md); they are reported x4 as clocks."""
import json
import os
import re
import sys

dst = sys.argv[1]
runs = [a.split("=", 1) for a in sys.argv[2:]]
# case -> (waves x ticks the counters are summed over, what it is)
CASES = {
    "stack_rollout256": (1024 * 256, "config 3, 16384 instances, on-device rollout of 256 ticks (team4v)"),
    "stack_resident": (1024 * 20000, "config 3, 16384 instances, resident ticks fed ahead (team4v)"),
    "stack_resident_state": (1024 * 20000, "config 3, resident ticks with the state kept by the kernel"),
    "pose_resident": (1024 * 20000, "config 2, 16384 instances, resident ticks fed ahead (quadv)"),
    "qp_rollout64": (256 * 64, "config 4, 16384 instances, on-device rollout of 64 ticks (lone-wave kernel)"),
    "qp_resident": (1020 * 10000, "config 4, 16320 instances, resident ticks fed ahead, every slot ANOTHER batch (stale hot starts)"),
}
out, rows = {}, []
for label, d in runs:
    for case, (wt, what) in CASES.items():
        raw = {}
        try:
            for line in open(os.path.join(d, "pmc_%s.txt" % case)):
                m = re.match(r"(\S+)\s+mean per dispatch ([0-9.]+) over (\d+) dispatches", line)
                if m:
                    raw[m.group(1)] = float(m.group(2))
        except OSError:
            continue
        if "SQ_INSTS_VALU" not in raw:
            continue
        per = {k: raw[k] / wt for k in raw}
        ent = {"what": what, "build": label, "wave_ticks": wt,
               "insts_per_tick_and_wave": {k[9:].lower(): round(per[k], 1) for k in per if k.startswith("SQ_INSTS_")},
               "clk_per_tick_and_wave": {k[3:].lower(): round(4 * per[k]) for k in
                                         ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")
                                         if k in per}}
        c = ent["clk_per_tick_and_wave"]
        if "wave_cycles" in c:
            ent["us_per_tick_at_2p4GHz"] = round(c["wave_cycles"] / 2400.0, 3)
        out["%s/%s" % (case, label)] = ent
        i = ent["insts_per_tick_and_wave"]
        rows.append("| %s | %s | %.0f | %.0f | %.0f | %d | %d | %d | %.2f |" % (
            case, label, i.get("valu", 0), i.get("salu", 0), i.get("valu", 0) + i.get("salu", 0) + i.get("branch", 0)
            + i.get("vmem_rd", 0) + i.get("vmem_wr", 0) + i.get("smem", 0),
            c.get("active_inst_any", 0), c.get("wait_any", 0), c.get("wave_cycles", 0), ent.get("us_per_tick_at_2p4GHz", 0)))
with open(dst + ".json", "w") as f:
    json.dump(out, f, indent=1)
with open(dst + ".md", "w") as f:
    f.write("# Round 6: the boundary-free tick kernels per tick and wave (PMC, `tools/profile_round6.sh`)\n\n"
            "One counter group per `rocprofv3 --pmc` run; counters of one dispatch divided by waves x ticks; cycle counters x 4.\n\n"
            "| kernel | build | VALU | SALU | all instructions | issue clocks | wait clocks | wave clocks | us per tick at 2.4 GHz |\n"
            "|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
    notes = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r6_boundary_free_reading.md")
    if os.path.exists(notes):
        f.write("\n" + open(notes).read())
print("\n".join(rows))
