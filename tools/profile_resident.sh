#!/bin/bash
# PMC passes of the resident tick kernel (one counter group per rocprofv3 --pmc run, no tracing):
#   gpurun -- bash tools/profile_resident.sh        -> gpurun_out/r3res/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3res
mkdir -p $OUT
export TMPDIR=/tmp
TICKS=20000
for MODE in plain state; do
    rm -f $OUT/pmc_resident_$MODE.txt
    for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
        rm -rf /tmp/pmc_res
        if [ $MODE = state ]; then
            timeout 120 rocprofv3 --pmc $C -d /tmp/pmc_res -- python3 tools/resident_once.py $TICKS state > $OUT/run_$MODE.log 2>> $OUT/pmc_$MODE.err
        else
            timeout 120 rocprofv3 --pmc $C -d /tmp/pmc_res -- python3 tools/resident_once.py $TICKS > $OUT/run_$MODE.log 2>> $OUT/pmc_$MODE.err
        fi
        echo "# $C   ($(tail -1 $OUT/run_$MODE.log))" >> $OUT/pmc_resident_$MODE.txt
        python3 tools/rocprof_counters.py /tmp/pmc_res resident >> $OUT/pmc_resident_$MODE.txt 2>&1
    done
    tail -30 $OUT/pmc_resident_$MODE.txt
done
