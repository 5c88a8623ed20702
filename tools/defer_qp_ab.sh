#!/bin/bash
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    line=$(env "$@" timeout 200 python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150 $BARGS 2>/dev/null | tail -1)
    python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-56s %-24s %.3f us/tick' % (sys.argv[2], d['config']['kernel'], d['ms_per_step']*1e3))" "$line" "$label" 2>/dev/null || echo "$label FAILED: ${line:0:200}"
}
for b in 16384 32768 131072 524288; do
    for hot in 0 1; do
    BARGS="--workload qp --batch $b --qp-hot $hot"
    [ $b -ge 500000 ] && BARGS="$BARGS --steps 200 --warmup 20 --replays 8"
    run "qp B=$b hot=$hot rows at once" CLIK_JIT_DEFINES=-DCLIK_DEFER_INPUT_ROWS=0 CLIK_QP_FOLIO=0
    run "qp B=$b hot=$hot input rows behind the robot_var rows" CLIK_QP_FOLIO=0
    done
done
