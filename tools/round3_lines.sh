#!/bin/bash
# The bench lines kept under profiles/r3_bench_lines.jsonl (one MI355X box), each preceded by its command:
#   gpurun -- bash tools/round3_lines.sh        -> gpurun_out/r3lines/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3lines
mkdir -p $OUT
: > $OUT/r3_bench_lines.jsonl
line () {   # bench args...
    echo "# python bench.py $*" >> $OUT/r3_bench_lines.jsonl
    python bench.py "$@" 2>> $OUT/err.log | grep '^{' | tail -1 >> $OUT/r3_bench_lines.jsonl
}
line
line --steps 20 --warmup 5
line --workload qp --cpu-baseline 0 --extras 0
line --workload qp --qp-lanes 4 --cpu-baseline 0 --extras 0
line --workload qp --qp-hot 1 --cpu-baseline 0 --extras 0
line --workload qp --batch 131072 --cpu-baseline 0 --extras 0
line --workload pose --cpu-baseline 0 --extras 0
line --batch 1048576 --steps 400 --warmup 40 --cpu-baseline 0 --extras 0
echo "# CLIK_JIT_DEFINES=-DCLIK_OCC3 python bench.py --batch 1048576 --steps 400 --warmup 40 --cpu-baseline 0 --extras 0   (experiment: three waves per SIMD for the lane kernel, 168 VGPRs + 624 B of scratch per lane)" >> $OUT/r3_bench_lines.jsonl
CLIK_JIT_DEFINES=-DCLIK_OCC3 python bench.py --batch 1048576 --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 2>> $OUT/err.log | grep '^{' | tail -1 >> $OUT/r3_bench_lines.jsonl
line --batch 32768 --cpu-baseline 0 --extras 0
line --ticks-per-launch 256 --steps 2048 --warmup 256 --cpu-baseline 0 --extras 0
line --workload qp --ticks-per-launch 64 --steps 2048 --warmup 256 --cpu-baseline 0 --extras 0
line --dist interior --cpu-baseline 0 --extras 0
echo "# CLIK_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --min-timed-ms 200 --allgather 1 --cpu-baseline 0   (two ranks on ONE GPU over gloo: a smoke test of the multi-rank code path, not a scaling number)" >> $OUT/r3_bench_lines.jsonl
CLIK_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --min-timed-ms 200 --allgather 1 --cpu-baseline 0 2>> $OUT/err.log | grep '^{' | tail -1 >> $OUT/r3_bench_lines.jsonl
echo "# CLIK_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --min-timed-ms 200 --global-batch 32768 --cpu-baseline 0   (same smoke test, strong scaling)" >> $OUT/r3_bench_lines.jsonl
CLIK_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --min-timed-ms 200 --global-batch 32768 --cpu-baseline 0 2>> $OUT/err.log | grep '^{' | tail -1 >> $OUT/r3_bench_lines.jsonl
python tools/moe_qp_bench.py > $OUT/r3_moe_qp_bench.txt 2>> $OUT/err.log
python tools/resident_probe.py > $OUT/r3_resident_probe.txt 2>> $OUT/err.log
python tools/notebook_bench.py > $OUT/r3_notebook_bench.txt 2>> $OUT/err.log
grep -c '^{' $OUT/r3_bench_lines.jsonl
