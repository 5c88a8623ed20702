#!/bin/bash
# Round 5: four lanes per instance with the sin / cos evaluations split over the quad (quadv / front4) against one lane
# per instance, and how the waves are packed into blocks.   tools/quad_ab.sh > gpurun_out/.../quad_ab.txt
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    line=$(env "$@" timeout 120 python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 500 --ramp-ms 150 $BARGS 2>/dev/null | tail -1)
    python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-64s %-28s %.3f us/tick' % (sys.argv[2], d['config']['kernel'], d['ms_per_step']*1e3))" "$line" "$label" 2>/dev/null || echo "$label FAILED: ${line:0:200}"
}
for b in 1024 4096 16384; do
    BARGS="--workload pose --batch $b"
    run "pose B=$b one lane per instance" CLIK_QUAD_FRONT=0
    run "pose B=$b four lanes, one wave per block" CLIK_NOOP=1
    run "pose B=$b four lanes, four waves per block" CLIK_QUAD_BLOCK=256
done
for b in 1024 4096 16384; do
    BARGS="--workload qp --qp-hot 1 --batch $b"
    run "qp hot B=$b one lane per instance" CLIK_QP_FRONT4=0
    run "qp hot B=$b four lanes (up to 1 wave per CU)" CLIK_QP_FRONT4=1
    run "qp hot B=$b four lanes (up to 4 waves per CU)" CLIK_QP_FRONT4=4
done
for b in 4096 16384; do
    BARGS="--workload qp --batch $b"
    run "qp cold B=$b folio (sin / cos shared through LDS)" CLIK_NOOP=1
    run "qp cold B=$b lone wave" CLIK_QP_FOLIO=0 CLIK_QP_FRONT4=0
    run "qp cold B=$b four lanes per instance (front4)" CLIK_QP_FOLIO=0 CLIK_QP_FRONT4=4
done
