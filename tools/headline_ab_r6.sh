#!/bin/bash
# Round 6 A/B on one box: where the value-specialised kernels take their constants from.  The sincos polynomial's 16
# constants as literals (two s_mov_b32 each) or from a constant-memory pool (two s_load_dwordx16 at kernel start); the
# team kernel's per-lane role constants from selects on the lane number or from a 4 x 8 table (four vector loads).
# Launched ticks (a kernel start invalidates the scalar and vector L1 caches: every CU fetches the lines again, every
# tick), the on-device rollout and the resident ticks (fetched once per launch).
#   gpurun -- bash tools/headline_ab_r6.sh      -> gpurun_out/r6ab/headline_ab.txt
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6ab
mkdir -p $OUT
B="--extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150"
line () {   # label, defines, bench args...
    label=$1; defs=$2; shift; shift
    us=$(CLIK_JIT_DEFINES="$defs" python bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us  %s' % (d['ms_per_step']*1e3, d['config']['kernel']))")
    echo "$label | ${defs:-defaults} | $us" | tee -a $OUT/headline_ab.txt
}
: > $OUT/headline_ab.txt
for rep in 1 2; do
for defs in "" "-DCLIK_SINCOS_POOL=0" "-DCLIK_ROLE_TABLE=0" "-DCLIK_SINCOS_POOL=0 -DCLIK_ROLE_TABLE=0"; do
    line "stack 16384 tick" "$defs"
    line "stack 16384 rollout256" "$defs" --ticks-per-launch 256 --steps 2560 --warmup 256
    line "pose 16384 tick" "$defs" --workload pose
    line "pose 4096 tick" "$defs" --workload pose --batch 4096
    line "qp 16384 hot tick" "$defs" --workload qp --qp-hot 1
    line "stack 131072 tick" "$defs" --batch 131072
done
done
