#!/bin/bash
# Round-6 PMC passes + kernel stats of the LAUNCHED ticks the bench line reports with a roofline (as tools/profile_round5.sh):
#   gpurun -- bash tools/profile_round6_launched.sh [only-first=0]    -> gpurun_out/r6prof_launched/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6prof_launched
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--cpu-baseline 0 --extras 0"
run_stats () {   # name, bench args...
    name=$1; shift
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python3 bench.py --steps 400 --warmup 50 --ramp-ms 50 --min-timed-ms 20 --replays 3 $COMMON "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_stats.err
    python3 tools/rocprof_summary.py /tmp/prof_$name $OUT/${name}_kernel_stats.csv > $OUT/${name}_stats.txt 2>&1
    tail -4 $OUT/${name}_stats.txt
}
run_pmc () {     # name, "counter list", bench args...
    name=$1; ctrs=$2; shift; shift
    rm -rf /tmp/pmc_$name
    rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -- python3 bench.py --steps 60 --warmup 10 --graph 0 --ramp-ms 5 --min-timed-ms 1 --replays 1 $COMMON "$@" > /dev/null 2> $OUT/pmc_${name}.err
    echo "# $ctrs" >> $OUT/pmc_${name}.txt
    python3 tools/rocprof_counters.py /tmp/pmc_$name solve_static >> $OUT/pmc_${name}.txt 2>&1
    tail -3 $OUT/pmc_${name}.txt
}
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE")
profile () {     # name, bench args...
    name=$1; shift
    run_stats $name "$@"
    rm -f $OUT/pmc_$name.txt
    for C in "${GROUPS_[@]}"; do
        run_pmc $name "$C" "$@"
    done
}
profile stack_team4v_16384
if [ "${1:-0}" = "1" ]; then ls $OUT; exit 0; fi
profile pose_quadv_4096 --workload pose --batch 4096
profile pose_quadv_16384 --workload pose
profile qp_16384_folio --workload qp
profile qp_16384_hot --workload qp --qp-hot 1
profile stack_lanev_131072 --batch 131072
profile qp_131072 --workload qp --batch 131072
profile qp_4096_folio --workload qp --batch 4096
profile stack_lanev_1M --batch 1048576
ls $OUT
