#!/bin/bash
# Copies what `gpurun -- bash tools/evidence_round6.sh` (+ `tools/profile_round6.sh shipped`) left under gpurun_out/ into
# profiles/r6_* (run here, after the GPU call).
set -eu
cd "$(dirname "$0")/.."
python tools/make_counters_json.py gpurun_out/r6prof_launched profiles/r6_counters.json > /dev/null
for f in gpurun_out/r6prof_launched/*_kernel_stats.csv; do cp "$f" "profiles/r6_$(basename "$f")"; done
cp gpurun_out/r6body/r6_body_time.json gpurun_out/r6body/r6_body_time.csv profiles/
cp gpurun_out/r6ev/stamp_body.txt profiles/r6_body_time.txt
cp gpurun_out/r6ev/moe_rollout_kernel_stats.csv profiles/r6_moe_rollout_kernel_stats.csv
cp gpurun_out/r6ev/notebook_bench.txt profiles/r6_notebook_bench.txt
cp gpurun_out/r6ev/bench_default_line.json profiles/r6_bench_default_line.json
cp gpurun_out/r6ev/bench_driver_like.json profiles/r6_bench_driver_like_line.json
cp gpurun_out/r6ev/bench_2ranks_shared_gpu.json profiles/r6_bench_2ranks_shared_gpu.json
cp gpurun_out/r6floor/floor.txt profiles/r6_floor_probe.txt
python - <<'PY'
import csv
rows = list(csv.reader(open("gpurun_out/r6ev/moe_rollout_kernel_stats.csv")))
by = {r[0].split("<")[0].replace("void clik::", ""): r for r in rows[1:]}
def per(name):
    r = by[name]
    return float(r[2]) / int(r[1]), int(r[1])
pr, n1 = per("pinv_rollout_static_kernel")
qr, n2 = per("qp_rollout_static_kernel")
out = open("gpurun_out/r6ev/moe_rollout.txt").read()
out += ("\nDEVICE time per tick (rocprofv3 --kernel-trace --stats of this run, profiles/r6_moe_rollout_kernel_stats.csv; "
        "%d + %d launches of 256 ticks each):\n" % (n1, n2))
out += "  pinv skills, pinv_rollout_static_kernel:  %.1f us per launch = %.2f us per tick   (a launch per tick under the trace: %s / %s us)\n" % (
    pr, pr / 256, by["pinv_solve_static_kernel"][3], by["pinv_solve_static_mp_kernel"][3])
out += "  QP wall skills, qp_rollout_static_kernel: %.1f us per launch = %.2f us per tick  (a launch per tick under the trace: %s us)\n" % (
    qr, qr / 256, by["qp_solve_static_kernel"][3])
open("profiles/r6_moe_rollout.txt", "w").write(out)
print(out[-600:])
PY
python tools/make_resident_counters.py profiles/r6_boundary_free_counters flags=gpurun_out/r6prof_flags diet=gpurun_out/r6prof_diet \
    ticket_early=gpurun_out/r6prof_final waits_placed=gpurun_out/r6prof_final2 rows_lds=gpurun_out/r6prof_final3 \
    shipped=gpurun_out/r6prof_shipped | grep shipped
