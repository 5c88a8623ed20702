#!/bin/bash
# Round 6 A/B on one box: the compiler's instruction scheduling strategy for the value-specialised kernels.  A lone wave
# pays 8.5 cycles for a DEPENDENT fp64 instruction against 4.1 for an independent one (profiles/r2_lanes_head_to_head.md),
# and the ticks are chains (FK joint by joint, LDL' pivots, substitutions): the default strategy schedules for register
# pressure / occupancy, `-mllvm -amdgpu-sched-strategy=max-ilp` for instruction-level parallelism.
#   gpurun -- bash tools/sched_ab_r6.sh      -> gpurun_out/r6sched/sched_ab.txt
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6sched
mkdir -p $OUT
B="--extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150"
line () {   # label, defines, bench args...
    label=$1; defs=$2; shift; shift
    us=$(CLIK_JIT_DEFINES="$defs" python bench.py $B "$@" 2>$OUT/last.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us  %s  check %s' % (d['ms_per_step']*1e3, d['config']['kernel'], d.get('check',{}).get('ok')))")
    echo "$label | ${defs:-defaults} | $us" | tee -a $OUT/sched_ab.txt
}
: > $OUT/sched_ab.txt
for rep in 1 2; do
for defs in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-sched-strategy=iterative-ilp"; do
    line "stack 16384 tick" "$defs"
    line "stack 16384 rollout256" "$defs" --ticks-per-launch 256 --steps 2560 --warmup 256
    line "pose 16384 tick" "$defs" --workload pose
    line "qp 16384 cold tick" "$defs" --workload qp
    line "qp 16384 hot tick (standing)" "$defs" --workload qp --qp-hot 2
    line "qp 16384 rollout64" "$defs" --workload qp --ticks-per-launch 64 --steps 640 --warmup 64
    line "stack 131072 tick" "$defs" --batch 131072
    line "qp 131072 tick" "$defs" --workload qp --batch 131072
done
done
