#!/bin/bash
# Round 6: does the launched config-3 tick still follow its instruction count?  The same kernel on fewer waves (4096 / 1024
# instances: 256 / 64 waves), on inputs that skip the cone test's second tier (interior: ~90 instructions fewer per wave),
# and config 2 beside it.     gpurun -- bash tools/floor_probe_r6.sh   -> gpurun_out/r6floor/floor.txt
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6floor
mkdir -p $OUT
B="--extras 0 --cpu-baseline 0 --min-timed-ms 400 --ramp-ms 150"
line () {
    label=$1; shift
    us=$(python bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us  %s' % (d['ms_per_step']*1e3, d['config']['kernel']))")
    echo "$label | $us" | tee -a $OUT/floor.txt
}
: > $OUT/floor.txt
for rep in 1 2; do
line "stack mixed 16384"
line "stack interior 16384" --dist interior
line "stack mixed 4096" --batch 4096
line "stack interior 4096" --batch 4096 --dist interior
line "stack mixed 1024" --batch 1024
line "stack mixed 64" --batch 64
line "stack mixed 16384 ring 1" --ring 1
line "stack mixed 16384 eager" --graph 0
line "pose mixed 16384" --workload pose
line "pose mixed 1024" --workload pose --batch 1024
done
