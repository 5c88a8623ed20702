#!/bin/bash
# Round 6: everything profiles/r6_* is made from, in ONE call on one MI355X box (so that the bench line, the PMC passes,
# the kernel stats and the body stamps come from the same build on the same device):
#   gpurun --timeout 3600 -- bash tools/evidence_round6.sh      -> gpurun_out/r6ev/, r6prof_launched/, r6body/, r6floor/
# then, here:  python tools/make_counters_json.py gpurun_out/r6prof_launched profiles/r6_counters.json ; cp ... profiles/
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6ev
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default_line.err
bash tools/profile_round6_launched.sh > $OUT/profile_launched.log 2>&1
python tools/stamp_body.py 300 --light > $OUT/stamp_body.txt 2>&1
rm -rf /tmp/prof_moe
rocprofv3 --kernel-trace --stats -d /tmp/prof_moe -- python3 tools/moe_rollout_bench.py > $OUT/moe_rollout.txt 2> $OUT/moe_rollout.err
python3 tools/rocprof_summary.py /tmp/prof_moe $OUT/moe_rollout_kernel_stats.csv > $OUT/moe_rollout_stats.txt 2>&1
python tools/notebook_bench.py > $OUT/notebook_bench.txt 2> $OUT/notebook_bench.err
bash tools/floor_probe_r6.sh > $OUT/floor.log 2>&1
CLIK_BENCH_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --cpu-baseline 0 > $OUT/bench_2ranks_shared_gpu.json 2> $OUT/bench_2ranks_shared_gpu.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
ls -la $OUT
