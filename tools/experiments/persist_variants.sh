#!/bin/bash
# Persistent lane kernel (CLIK_LANE_PERSIST_MIN) and -disable-machine-licm on the run-time instantiated kernels: ticks
# of BASELINE config 3 / 2 / 4 per batch size under the four combinations.   bash tools/persist_variants.sh [out]
out=${1:-gpurun_out/r4persist}
mkdir -p $out
LICM="-mllvm -disable-machine-licm"
PERSIST="-DCLIK_LANE_PERSIST"     # (the persistent kernel is compiled only with this define)
line() {   # name, env..., -- bench args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --extras 0 --cpu-baseline 0 --min-timed-ms 400 "$@" > $out/$name.json 2> $out/$name.err
  python - "$name" $out/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-34s %9.3f us/tick  %s" % (sys.argv[1], d["ms_per_step"] * 1e3, d["config"].get("kernel")))
except Exception as e:
    print("%-34s FAILED %s" % (sys.argv[1], e))
PY
}
CLIK_JIT_DEFINES="$PERSIST" python tools/persist_check.py --both 1048613 > $out/check_plain_flags.txt 2>&1; tail -3 $out/check_plain_flags.txt
CLIK_JIT_DEFINES="$PERSIST $LICM" python tools/persist_check.py --both 1048613 > $out/check_licm.txt 2>&1; tail -3 $out/check_licm.txt
for B in 131072 262144 524288 1048576; do
  line stack_${B}_plain            CLIK_LANE_PERSIST_MIN=9999999999 -- --batch $B
  line stack_${B}_persist          CLIK_LANE_PERSIST_MIN=1 "CLIK_JIT_DEFINES=$PERSIST" -- --batch $B
  line stack_${B}_licm             CLIK_LANE_PERSIST_MIN=9999999999 "CLIK_JIT_DEFINES=$LICM" -- --batch $B
  line stack_${B}_persist_licm     CLIK_LANE_PERSIST_MIN=1 "CLIK_JIT_DEFINES=$PERSIST $LICM" -- --batch $B
done
line stack_16384_plain   X=1 -- --batch 16384
line stack_16384_licm    "CLIK_JIT_DEFINES=$LICM" -- --batch 16384
line pose_16384_plain    X=1 -- --batch 16384 --workload pose
line pose_16384_licm     "CLIK_JIT_DEFINES=$LICM" -- --batch 16384 --workload pose
line pose_1M_plain       CLIK_LANE_PERSIST_MIN=9999999999 -- --batch 1048576 --workload pose
line pose_1M_persist_licm CLIK_LANE_PERSIST_MIN=1 "CLIK_JIT_DEFINES=$PERSIST $LICM" -- --batch 1048576 --workload pose
line qp_16384_plain      X=1 -- --batch 16384 --workload qp
line qp_16384_licm       "CLIK_JIT_DEFINES=$LICM" -- --batch 16384 --workload qp
line qp_131072_plain     X=1 -- --batch 131072 --workload qp
line qp_131072_licm      "CLIK_JIT_DEFINES=$LICM" -- --batch 131072 --workload qp
