#!/bin/bash
# The bench lines kept under profiles/r4_bench_lines.jsonl (one MI355X box, end of round 4), each preceded by its command;
# `--full 1`: the long form (every entry with its roofline, counters source, cpu_baseline).
#   gpurun -- bash tools/round4_lines.sh        -> gpurun_out/r4lines/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4lines
mkdir -p $OUT
: > $OUT/r4_bench_lines.jsonl
line () {   # [ENV=..]... bench args...
    envs=()
    while [[ "${1:-}" == *=* && "${1:-}" != --* ]]; do envs+=("$1"); shift; done
    echo "# ${envs[*]} python bench.py $*" >> $OUT/r4_bench_lines.jsonl
    env "${envs[@]}" python bench.py "$@" 2>> $OUT/err.log | grep '^{' | tail -1 >> $OUT/r4_bench_lines.jsonl
}
line
line --full 1
line --steps 20 --warmup 5
line --workload qp --cpu-baseline 0 --extras 0
line CLIK_QP_FOLIO=0 --workload qp --cpu-baseline 0 --extras 0
for sd in 1 2 3 4 5; do
  line --workload qp --seed $sd --cpu-baseline 0 --extras 0
  line CLIK_QP_FOLIO=0 --workload qp --seed $sd --cpu-baseline 0 --extras 0
done
line --workload qp --batch 4096 --cpu-baseline 0 --extras 0
line CLIK_QP_FOLIO=0 --workload qp --batch 4096 --cpu-baseline 0 --extras 0
line --workload qp --qp-hot 1 --cpu-baseline 0 --extras 0
line --workload qp --batch 131072 --cpu-baseline 0 --extras 0
line --workload pose --cpu-baseline 0 --extras 0
line --workload pose --batch 4096 --cpu-baseline 0 --extras 0
line --batch 131072 --cpu-baseline 0 --extras 0
line --batch 1048576 --steps 400 --warmup 40 --cpu-baseline 0 --extras 0
line --ticks-per-launch 256 --steps 2048 --warmup 256 --cpu-baseline 0 --extras 0
line --workload qp --ticks-per-launch 64 --steps 2048 --warmup 256 --cpu-baseline 0 --extras 0
line --dist interior --cpu-baseline 0 --extras 0
python tools/moe_qp_bench.py > $OUT/r4_moe_qp_bench.txt 2>> $OUT/err.log
grep -c '^{' $OUT/r4_bench_lines.jsonl
