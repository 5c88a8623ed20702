#!/bin/bash
# Round-6 PMC passes of the BOUNDARY-FREE tick kernels: the on-device rollouts and the resident
# tick kernels of both controllers, instructions and wait cycles PER TICK - each counter group in its own
# rocprofv3 --pmc run (never with tracing).
#   gpurun -- bash tools/profile_round6.sh [tag]       -> gpurun_out/r6prof[_tag]/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6prof${1:+_$1}
mkdir -p $OUT
export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE")
pmc () {     # name, kernel-name filter, command...
    name=$1; filt=$2; shift; shift
    rm -f $OUT/pmc_$name.txt
    for C in "${GROUPS_[@]}"; do
        rm -rf /tmp/pmc_$name
        timeout 180 rocprofv3 --pmc $C -d /tmp/pmc_$name -- "$@" > $OUT/run_$name.log 2>> $OUT/pmc_$name.err
        echo "# $C   ($(tail -1 $OUT/run_$name.log | cut -c1-160))" >> $OUT/pmc_$name.txt
        python3 tools/rocprof_counters.py /tmp/pmc_$name $filt >> $OUT/pmc_$name.txt 2>&1
    done
    tail -40 $OUT/pmc_$name.txt
}
ROLL="--graph 0 --ramp-ms 5 --min-timed-ms 1 --replays 1 --cpu-baseline 0 --extras 0"
pmc stack_rollout256 rollout python3 bench.py --ticks-per-launch 256 --steps 2560 --warmup 256 $ROLL
pmc stack_resident resident python3 tools/resident_once.py 20000
pmc stack_resident_state resident python3 tools/resident_once.py 20000 state
pmc pose_resident resident python3 tools/resident_once.py 20000 pose
pmc qp_rollout64 rollout python3 bench.py --workload qp --ticks-per-launch 64 --steps 640 --warmup 64 $ROLL
pmc qp_resident resident python3 tools/resident_once.py 10000 qp
ls $OUT
