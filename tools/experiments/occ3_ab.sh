#!/bin/bash
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    line=$(env "$@" timeout 200 python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150 $BARGS 2>/dev/null | tail -1)
    python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-56s %-24s %.3f us/tick' % (sys.argv[2], d['config']['kernel'], d['ms_per_step']*1e3))" "$line" "$label" 2>/dev/null || echo "$label FAILED: ${line:0:200}"
}
for b in 131072 262144 1048576; do
    BARGS="--workload stack --batch $b --steps 200 --warmup 20 --replays 8"
    run "stack B=$b shipped (2 waves per SIMD)" CLIK_NOOP=1
    run "stack B=$b held to 3 waves per SIMD" CLIK_JIT_DEFINES=-DCLIK_OCC3
done
