#!/bin/bash
# Round 5: the input_var rows requested behind the robot_var rows (CLIK_DEFER_INPUT_ROWS) against both at once.
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    line=$(env "$@" timeout 200 python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150 $BARGS 2>/dev/null | tail -1)
    python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-56s %-24s %.3f us/tick' % (sys.argv[2], d['config']['kernel'], d['ms_per_step']*1e3))" "$line" "$label" 2>/dev/null || echo "$label FAILED: ${line:0:200}"
}
for b in 32768 65536 131072 262144 1048576; do
    BARGS="--workload stack --batch $b"
    [ $b -ge 1000000 ] && BARGS="$BARGS --steps 200 --warmup 20 --replays 8"
    run "stack B=$b rows at once" CLIK_JIT_DEFINES=-DCLIK_DEFER_INPUT_ROWS=0
    run "stack B=$b input rows behind the robot_var rows" CLIK_NOOP=1
done
for b in 65536 131072; do
    BARGS="--workload pose --batch $b"
    run "pose B=$b rows at once" CLIK_JIT_DEFINES=-DCLIK_DEFER_INPUT_ROWS=0
    run "pose B=$b input rows behind the robot_var rows" CLIK_NOOP=1
done
