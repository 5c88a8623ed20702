#!/bin/bash
# VERDICT r3 item 6 ("measure, don't argue"): the headline tick (BASELINE config 3, 16384 instances, inputs "mixed") through
#   shipped      four lanes per instance, every lane evaluates FK / task rows / Gram itself (team4v)
#   front_once   -DCLIK_TEAM_FRONT_ONCE: FK and the task rows by lane 0 of the quad only, DPP broadcast to the other three
#   one_lane     --lanes 1: one lane per instance (lanev), the other end of the bracket "two instances per quad" sits in
# each under bench.py's bracket (same clock ramp, graphs of 2000 ticks), then the body time of the shipped kernels with
# LIGHT stamps (tools/stamp_body.py --light).      gpurun -- bash tools/team_variants.sh   -> gpurun_out/r4variants/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4variants
mkdir -p $OUT
COMMON="--extras 0 --cpu-baseline 0 --min-timed-ms 1000"
python3 bench.py $COMMON > $OUT/shipped.json 2> $OUT/shipped.err
CLIK_JIT_DEFINES="-DCLIK_TEAM_FRONT_ONCE" python3 bench.py $COMMON > $OUT/front_once.json 2> $OUT/front_once.err
python3 bench.py $COMMON --lanes 1 > $OUT/one_lane.json 2> $OUT/one_lane.err
python3 bench.py $COMMON --batch 32768 > $OUT/shipped_B32768.json 2> $OUT/shipped_B32768.err
python3 bench.py $COMMON --batch 32768 --lanes 4 > $OUT/team4_B32768.json 2> $OUT/team4_B32768.err
python3 tools/stamp_body.py 400 --light > $OUT/body_light.log 2>&1
cp gpurun_out/r4body/r4_body_time_light.* $OUT/ 2>/dev/null
for f in shipped front_once one_lane shipped_B32768 team4_B32768; do python3 -c "
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('%-16s %.3f us per tick  %s' % ('$f', d['ms_per_step']*1e3, d['config']['kernel']))"; done
tail -6 $OUT/body_light.log
