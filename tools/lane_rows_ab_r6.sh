#!/bin/bash
# Round 6 A/B on one box: the lane-per-instance value kernels with their rows copied memory -> LDS in whole 16-byte pieces
# (and stored back the same way) against rows loaded and stored lane by lane (-DCLIK_LANE_ROWS_LDS=0: 2 n strided 8-byte
# requests per lane, each instruction touching ~56 cache lines).
#   gpurun -- bash tools/lane_rows_ab_r6.sh      -> gpurun_out/r6lanerows/lane_rows_ab.txt
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6lanerows
mkdir -p $OUT
B="--extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150"
line () {   # label, defines, bench args...
    label=$1; defs=$2; shift; shift
    us=$(CLIK_JIT_DEFINES="$defs" python bench.py $B "$@" 2>$OUT/last.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us  %s  check %s' % (d['ms_per_step']*1e3, d['config']['kernel'], d.get('check',{}).get('ok')))")
    echo "$label | ${defs:-defaults (rows through LDS)} | $us" | tee -a $OUT/lane_rows_ab.txt
}
: > $OUT/lane_rows_ab.txt
for rep in 1 2; do
for defs in "" "-DCLIK_LANE_ROWS_LDS=0"; do
    line "stack 32768 tick" "$defs" --batch 32768
    line "stack 131072 tick" "$defs" --batch 131072
    line "stack 1M tick" "$defs" --batch 1048576 --steps 200 --warmup 20
    line "pose 131072 tick" "$defs" --workload pose --batch 131072
    line "qp 16384 hot tick (standing)" "$defs" --workload qp --qp-hot 2
    line "qp 16384 hot tick (moving)" "$defs" --workload qp --qp-hot 1
    line "qp 131072 tick" "$defs" --workload qp --batch 131072
done
done
