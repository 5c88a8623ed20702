#!/bin/bash
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    line=$(env "$@" timeout 200 python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150 $BARGS 2>/dev/null | tail -1)
    python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-56s %-24s %.3f us/tick' % (sys.argv[2], d['config']['kernel'], d['ms_per_step']*1e3))" "$line" "$label" 2>/dev/null || echo "$label FAILED: ${line:0:200}"
}
for rep in 1 2 3; do
for b in 131072 1048576; do
    BARGS="--workload stack --batch $b"
    [ $b -ge 1000000 ] && BARGS="$BARGS --steps 200 --warmup 20 --replays 8"
    run "rep $rep stack B=$b rows at once" CLIK_JIT_DEFINES=-DCLIK_DEFER_INPUT_ROWS=0
    run "rep $rep stack B=$b input rows deferred" CLIK_NOOP=1
done
done
