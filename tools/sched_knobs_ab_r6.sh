#!/bin/bash
# Round 6 A/B on one box, on top of the shipped scheduling strategies (casclik_amd/jit.py::sched_strategy): the scheduler's
# other knobs.    gpurun -- bash tools/sched_knobs_ab_r6.sh      -> gpurun_out/r6sched/sched_knobs_ab.txt
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6sched
mkdir -p $OUT
B="--extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150"
line () {   # label, defines, bench args...
    label=$1; defs=$2; shift; shift
    us=$(CLIK_JIT_DEFINES="$defs" python bench.py $B "$@" 2>$OUT/last.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us  %s  check %s' % (d['ms_per_step']*1e3, d['config']['kernel'], d.get('check',{}).get('ok')))")
    echo "$label | ${defs:-shipped} | $us" | tee -a $OUT/sched_knobs_ab.txt
}
: > $OUT/sched_knobs_ab.txt
for rep in 1 2; do
for defs in "" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-schedule-relaxed-occupancy" "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule" "-mllvm -amdgpu-use-amdgpu-trackers"; do
    line "stack 16384 tick" "$defs"
    line "pose 16384 tick" "$defs" --workload pose
    line "qp 16384 cold tick" "$defs" --workload qp
    line "qp 16384 hot tick (standing)" "$defs" --workload qp --qp-hot 2
    line "qp 16384 rollout64" "$defs" --workload qp --ticks-per-launch 64 --steps 640 --warmup 64
    line "stack 131072 tick" "$defs" --batch 131072
done
done
