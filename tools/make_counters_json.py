#!/usr/bin/env python3
"""profiles/r2_counters.json from the PMC passes of tools/profile_round.sh (gpurun_out/r2prof/pmc_*.txt):
raw per-dispatch means of every counter, and the derived figures bench.py prints (HBM traffic with the
gfx950 FETCH_SIZE x2 correction, executed fp64 flops, per-wave instruction counts, SQ busy / wait shares).
SQ_*_CYCLES-type counters tick once per 4 clocks (MI355X_MICROARCH.md); they are reported x4.
    python tools/make_counters_json.py gpurun_out/r2prof profiles/r2_counters.json
"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
CASES_R3 = {  # round 3 (tools/profile_round3.sh): large batches too
    "stack_team4v_16384": ("stack_mixed_B16384_kStackIiwa/team4v", "kStackIiwa/team4v", 172, 16384),
    "stack_lanev_131072": ("stack_mixed_B131072_kStackIiwa/lanev", "kStackIiwa/lanev", 172, 131072),
    "stack_lanev_1M": ("stack_mixed_B1048576_kStackIiwa/lanev", "kStackIiwa/lanev", 172, 1048576),
    "qp_16384": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v", "qp_static_kQpPoseIiwa/v", 220, 16384),
    "qp_131072": ("qp_mixed_B131072_qp_static_kQpPoseIiwa/v", "qp_static_kQpPoseIiwa/v", 220, 131072),
    "pose_lanev_16384": ("pose_mixed_B16384_kPose6Iiwa/lanev", "kPose6Iiwa/lanev", 172, 16384),
}
CASES_R4 = {  # round 4 (tools/profile_round4.sh): the hot-started QP tick gets its own passes
    "stack_team4v_16384": ("stack_mixed_B16384_kStackIiwa/team4v", "kStackIiwa/team4v", 172, 16384),
    "qp_16384": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v", "qp_static_kQpPoseIiwa/v", 220, 16384),
    "qp_16384_hot": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v_hot", "qp_static_kQpPoseIiwa/v", 220, 16384),
    "pose_lanev_16384": ("pose_mixed_B16384_kPose6Iiwa/lanev", "kPose6Iiwa/lanev", 172, 16384),
    "pose_lanev_4096": ("pose_mixed_B4096_kPose6Iiwa/lanev", "kPose6Iiwa/lanev", 172, 4096),
    "stack_lanev_131072": ("stack_mixed_B131072_kStackIiwa/lanev", "kStackIiwa/lanev", 172, 131072),
    "qp_131072": ("qp_mixed_B131072_qp_static_kQpPoseIiwa/v", "qp_static_kQpPoseIiwa/v", 220, 131072),
    # (tools/profile_round4b.sh: cold ticks up to one block per CU run four waves per 64 instances)
    "qp_16384_folio": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v/folio4", "qp_static_kQpPoseIiwa/v/folio4", 220, 16384),
    "qp_4096_folio": ("qp_mixed_B4096_qp_static_kQpPoseIiwa/v/folio4", "qp_static_kQpPoseIiwa/v/folio4", 220, 4096),
}
CASES_R5 = {  # round 5 (tools/profile_round5.sh): config 2 runs four lanes per instance (quadv), FOLIO shares its sin / cos
    "stack_team4v_16384": ("stack_mixed_B16384_kStackIiwa/team4v", "kStackIiwa/team4v", 172, 16384),
    "pose_quadv_4096": ("pose_mixed_B4096_kPose6Iiwa/quadv", "kPose6Iiwa/quadv", 172, 4096),
    "pose_quadv_16384": ("pose_mixed_B16384_kPose6Iiwa/quadv", "kPose6Iiwa/quadv", 172, 16384),
    "qp_16384_folio": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v/folio4", "qp_static_kQpPoseIiwa/v/folio4", 220, 16384),
    "qp_4096_folio": ("qp_mixed_B4096_qp_static_kQpPoseIiwa/v/folio4", "qp_static_kQpPoseIiwa/v/folio4", 220, 4096),
    "qp_16384_hot": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v_hot", "qp_static_kQpPoseIiwa/v", 220, 16384),
    "stack_lanev_131072": ("stack_mixed_B131072_kStackIiwa/lanev", "kStackIiwa/lanev", 172, 131072),
    "qp_131072": ("qp_mixed_B131072_qp_static_kQpPoseIiwa/v", "qp_static_kQpPoseIiwa/v", 220, 131072),
}
CASES = {  # pmc file tag -> (bench key, kernel label, algorithmic bytes per instance, batch)
    "stack_team4v": ("stack_mixed_B16384_kStackIiwa/team4v", "kStackIiwa/team4v", 172, 16384),
    "stack_team4": ("stack_mixed_B16384_kStackIiwa/team4", "kStackIiwa/team4", 172, 16384),
    "stack_lane": ("stack_mixed_B16384_kStackIiwa/mp2", "kStackIiwa/mp2", 172, 16384),
    "qp": ("qp_mixed_B16384_qp_static_kQpPoseIiwa/v", "qp_static_kQpPoseIiwa/v", 168, 16384),
    "pose_lanev": ("pose_mixed_B16384_kPose6Iiwa/lanev", "kPose6Iiwa/lanev", 172, 16384),
}
CASES_R6 = dict(CASES_R5)        # round 6 (tools/profile_round6_launched.sh): the same configurations on the round's kernels
CASES_R6["stack_lanev_1M"] = ("stack_mixed_B1048576_kStackIiwa/lanev", "kStackIiwa/lanev", 172, 1048576)
if "r3" in dst:
    CASES = CASES_R3
if "r4" in dst:
    CASES = CASES_R4
if "r5" in dst:
    CASES = CASES_R5
if "r6" in dst:
    CASES = CASES_R6
out = {"note3": "round 3: the same passes (tools/profile_round3.sh, 310 dispatches each) incl. 131072 and 1 M instances; "
                "valu_issue_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the "
                "kernel's duration in which a SIMD issues a VALU instruction, averaged over all SIMDs; "
                "fp64_share_of_valu = fp64 FMA + MUL + ADD instructions / all VALU instructions.",
       "note": "rocprofv3 --pmc passes (one counter group per run, no tracing) of python3 bench.py --graph 0 at 16384 "
               "instances, inputs 'mixed'; mean per dispatch over 520 dispatches, summed over XCDs "
               "(tools/rocprof_counters.py).  traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the x2 is the "
               "gfx950 correction calibrated in profiles/r1_traffic.json).  fp64_flops_per_launch = 64 lanes x "
               "(2 FMA + MUL + ADD) wave-instructions EXECUTED (the team kernel's four lanes per instance repeat the "
               "front end, so its executed flops exceed the lane kernels').  insts_per_wave = counter / SQ_WAVES; "
               "cycle-type SQ counters are in units of 4 clocks and are listed x4 as clk_per_wave."}
for tag, (key, kernel, bpi, B) in CASES.items():
    raw = {}
    try:
        for line in open("%s/pmc_%s.txt" % (src, tag)):
            m = re.match(r"(\S+)\s+mean per dispatch ([0-9.]+) over (\d+) dispatches", line)
            if m:
                raw[m.group(1)] = float(m.group(2))
    except OSError:
        continue
    if not raw:
        continue
    w = raw.get("SQ_WAVES", 0.0) or 1.0
    ent = {"kernel": kernel, "raw_per_dispatch": raw, "algorithmic_bytes": bpi * B,
           "traffic_bytes": int(round(2 * raw["FETCH_SIZE"] * 1024 + raw["WRITE_SIZE"] * 1024)),
           "fp64_flops_per_launch": 64.0 * (2 * raw["SQ_INSTS_VALU_FMA_F64"] + raw["SQ_INSTS_VALU_MUL_F64"]
                                            + raw["SQ_INSTS_VALU_ADD_F64"]),
           "waves": w,
           "insts_per_wave": {k[9:].lower(): round(raw[k] / w, 1) for k in raw if k.startswith("SQ_INSTS_")},
           "clk_per_wave": {k[3:].lower(): round(4 * raw[k] / w) for k in
                            ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
                             "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS") if k in raw}}
    # (GRBM_GUI_ACTIVE spans the profiler's serialised dispatch, not only the kernel: a usable denominator only
    # where the kernel is long against that overhead: 70 us at 1 M instances against ~8 us of gaps)
    if "GRBM_GUI_ACTIVE" in raw and "SQ_ACTIVE_INST_VALU" in raw and B >= 1048576:
        ent["gpu_active_clk"] = raw["GRBM_GUI_ACTIVE"] / 8.0
        ent["valu_issue_frac"] = 4.0 * raw["SQ_ACTIVE_INST_VALU"] / (1024.0 * raw["GRBM_GUI_ACTIVE"] / 8.0)
        ent["fp64_share_of_valu"] = (raw["SQ_INSTS_VALU_FMA_F64"] + raw["SQ_INSTS_VALU_MUL_F64"]
                                     + raw["SQ_INSTS_VALU_ADD_F64"]) / raw["SQ_INSTS_VALU"]
    out[key] = ent
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, "traffic", v["traffic_bytes"], "flops %.3g" % v["fp64_flops_per_launch"], v["insts_per_wave"], v["clk_per_wave"])
