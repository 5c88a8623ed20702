#!/bin/bash
# The measurements behind profiles/r4_qp_wave_portfolio.txt (one MI355X): the four-waves-per-64-instances QP kernel (FOLIO,
# clik_qp_static.hpp) against the lone-wave kernel - parity / determinism, ticks over batch sizes and input seeds, the cost
# of the arrangement alone (four identical waves; three of them leaving at once; one or two waves per block), per-wave stamps.
#   gpurun -- bash tools/qp_folio_variants.sh [out = gpurun_out/r4folio_all]
out=${1:-gpurun_out/r4folio_all}
mkdir -p $out
tick () {   # label, env..., -- bench args
  label=$1; shift; envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 500 --workload qp "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-58s %.3f us/tick' % ('$label', d['ms_per_step']*1e3))"
}
{
echo "## tools/qp_folio_check.py 16384"; python tools/qp_folio_check.py 16384 2>&1 | grep -v amdgpu.ids | tail -5
echo "## tools/qp_folio_check.py 4096 --delays"; python tools/qp_folio_check.py 4096 --delays 2>&1 | grep "bit-equal"
echo "## batch sizes (seed 0): lone-wave kernel | four waves, different starts | four waves, the lone-wave kernel's start"
for B in 256 1024 2048 4096 8192 12288 16384; do
  tick "B $B lone" CLIK_QP_FOLIO=0 -- --batch $B
  tick "B $B four waves" CLIK_QP_FOLIO=1 -- --batch $B
  tick "B $B four identical waves" CLIK_QP_FOLIO=1 CLIK_QP_FOLIO_SAME=1 -- --batch $B
done
echo "## input seeds at 16384 instances"
for sd in 0 1 2 3 4 5; do
  tick "seed $sd lone" CLIK_QP_FOLIO=0 -- --batch 16384 --seed $sd
  tick "seed $sd four waves" CLIK_QP_FOLIO=1 -- --batch 16384 --seed $sd
done
echo "## waves per block at 16384 (1: the bookkeeping alone; 2: relaxed x 12 | reverse relaxed x 3), and four identical waves with three leaving at once"
for W in 1 2; do
  tick "W $W identical" "CLIK_JIT_DEFINES=-DCLIK_QP_FOLIO_WAVES=$W" CLIK_QP_FOLIO=1 CLIK_QP_FOLIO_SAME=1 -- --batch 16384
  tick "W $W different starts" "CLIK_JIT_DEFINES=-DCLIK_QP_FOLIO_WAVES=$W" CLIK_QP_FOLIO=1 -- --batch 16384
done
tick "four identical waves, three leave at once" "CLIK_JIT_DEFINES=-DCLIK_QP_FOLIO_IDLE" CLIK_QP_FOLIO=1 CLIK_QP_FOLIO_SAME=1 -- --batch 16384
echo "## per-wave stamps (tools/stamp_folio.py)"
for f in "0 0" "1 1" "1 0"; do set -- $f; CLIK_QP_FOLIO=$1 CLIK_QP_FOLIO_SAME=$2 python tools/stamp_folio.py 16384 200 2>&1 | grep FOLIO; done
} 2>&1 | tee $out/qp_folio_variants.txt
