#!/bin/bash
# Round profiles on the GPU box: rocprofv3 kernel stats of the bench lines and PMC counter passes
# (each --pmc set in its own run, never with tracing) -> gpurun_out/r2prof/*.txt ; the summaries that are
# judged get copied into profiles/ by hand.
#   gpurun -- bash tools/profile_round.sh
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
STEPS="--steps 400 --warmup 50 --cpu-baseline 0 --ramp-ms 50 --min-timed-ms 20 --replays 3"
run_stats () {   # name, bench args...
    name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python3 bench.py $STEPS "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_stats.err
    python3 tools/rocprof_summary.py /tmp/prof_$name $OUT/${name}_kernel_stats.csv > $OUT/${name}_stats.txt 2>&1
    cat $OUT/${name}_stats.txt | tail -4
}
run_pmc () {     # name, "counter list", bench args...
    name=$1; ctrs=$2; shift; shift
    rm -rf /tmp/pmc_$name
    rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -- python3 bench.py --steps 100 --warmup 20 --cpu-baseline 0 --graph 0 --ramp-ms 5 --min-timed-ms 1 --replays 1 "$@" > /dev/null 2> $OUT/pmc_${name}.err
    python3 tools/rocprof_counters.py /tmp/pmc_$name solve_static >> $OUT/pmc_${name}.txt 2>&1
    tail -3 $OUT/pmc_${name}.txt
}
for V in team4v:0:1 team4:0:0 lane:1:1; do
    tag=${V%%:*}; rest=${V#*:}; lanes=${rest%%:*}; export CLIK_JIT_VALUES=${rest##*:}
    run_stats stack_$tag --lanes $lanes
    rm -f $OUT/pmc_stack_$tag.txt
    for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
             "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
        run_pmc stack_$tag "$C" --lanes $lanes
    done
done
unset CLIK_JIT_VALUES
run_stats pose_lanev --workload pose
rm -f $OUT/pmc_pose_lanev.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    run_pmc pose_lanev "$C" --workload pose
done
run_stats qp --workload qp
rm -f $OUT/pmc_qp.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"; do
    run_pmc qp "$C" --workload qp
done
ls $OUT
