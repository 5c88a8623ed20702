#!/bin/bash
# Round 5: block shape and kernel-argument preload of the headline kernel (and config 2).  tools/headline_ab.sh
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    line=$(env "$@" timeout 120 python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 600 --ramp-ms 150 $BARGS 2>/dev/null | tail -1)
    python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-64s %-28s %.3f us/tick' % (sys.argv[2], d['config']['kernel'], d['ms_per_step']*1e3))" "$line" "$label" 2>/dev/null || echo "$label FAILED: ${line:0:200}"
}
for wl in "stack" "pose" "pose --batch 4096" "qp" "qp --qp-hot 1"; do
    BARGS="--workload $wl"
    run "$wl B=16384 shipped (14 kernarg dwords preloaded)" CLIK_NOOP=1
    run "$wl B=16384 no kernarg preload" CLIK_JIT_KERNARG_PRELOAD=0
    run "$wl B=16384 shipped again" CLIK_NOOP=1
done
