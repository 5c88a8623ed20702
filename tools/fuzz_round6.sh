#!/bin/bash
# Round-6 randomised parity sweeps on the GPU box, every tool under the ONE stated rule (tests/tolerances.py).
#   bash tools/fuzz_round6.sh [out = gpurun_out/r6fuzz]   ->  <out>/r6_fuzz_summary.txt (+ the full logs beside it)
out=${1:-gpurun_out/r6fuzz}
mkdir -p $out
run() { name=$1; shift; python "$@" > $out/$name.log 2>&1; echo "$name rc $?"; }
run fuzz_parity_40_251   tools/fuzz_parity.py 40 251
run fuzz_parity_60_231   tools/fuzz_parity.py 60 231
run fuzz_team_30_207      tools/fuzz_team.py 30 207
run fuzz_team_20_203      tools/fuzz_team.py 20 203
run fuzz_qp_box_16_209    tools/fuzz_qp_box.py 16 209
run fuzz_qp_mixed_30_213 tools/fuzz_qp_mixed.py 30 213
run fuzz_qp_dynamic_200_205 tools/fuzz_qp_dynamic.py 200 205
run fuzz_qp_wide_40_200 tools/fuzz_qp_wide.py 40 200 96
FUZZ_ANGLES=1 python tools/fuzz_parity.py 40 277 > $out/fuzz_parity_angles_40_277.log 2>&1; echo "fuzz_parity_angles rc $?"
s=$out/r6_fuzz_summary.txt
{
echo "# Round 6 randomised parity sweeps on one MI355X (final kernels; every tool holds every instance to the stated rule"
echo "# err <= max(1e-12, 8 u kappa) of tests/tolerances.py; full logs are scratch under $out)"
for f in fuzz_parity_40_251 fuzz_parity_60_231; do
  echo; echo "## tools/fuzz_parity.py  (log $f)"
  grep -c "MISMATCH" $out/$f.log | sed 's/^/instances beyond the rule (MISMATCH lines): /'
  grep -o "([0-9.]* x tol)" $out/$f.log | tr -d '(' | sort -g | tail -1 | sed 's/^/worst err \/ tol of a pinv skill: /'
  tail -2 $out/$f.log
done
for f in fuzz_team_30_207 fuzz_team_20_203; do
  echo; echo "## tools/fuzz_team.py  (log $f)"
  grep -o "([0-9.]* x tol" $out/$f.log | tr -d '(' | sort -g | tail -1 | sed 's/^/worst err \/ tol: /'
  tail -1 $out/$f.log
done
echo; echo "## tools/fuzz_qp_box.py  (log fuzz_qp_box_16_209)"; tail -1 $out/fuzz_qp_box_16_209.log
echo; echo "## tools/fuzz_qp_mixed.py  (log fuzz_qp_mixed_30_213)"; grep "skipped" $out/fuzz_qp_mixed_30_213.log | cut -c1-200; tail -1 $out/fuzz_qp_mixed_30_213.log
echo; echo "## tools/fuzz_qp_dynamic.py  (log fuzz_qp_dynamic_200_205)"; tail -1 $out/fuzz_qp_dynamic_200_205.log
echo; echo "## tools/fuzz_qp_wide.py  (log fuzz_qp_wide_40_200)"; tail -1 $out/fuzz_qp_wide_40_200.log
echo; echo "## FUZZ_ANGLES=1 tools/fuzz_parity.py 40 277  (generated constraints also draw atan2 / asin / acos / atan / tanh / fmin / fmax)"
echo "instances beyond the rule (MISMATCH lines): $(grep -c MISMATCH $out/fuzz_parity_angles_40_277.log); skills whose constraints ran as generated device code: $(grep -c 'pinv dynamic refused: the skill has constraint expressions' $out/fuzz_parity_angles_40_277.log) of 40"; tail -2 $out/fuzz_parity_angles_40_277.log
echo; echo "## tools/fuzz_qp_mixed.py, every skill of the sweep"; grep -E "^ *[0-9]+ (ur5|iiwa)" $out/fuzz_qp_mixed_30_213.log | cut -c1-200
echo; echo "## tools/fuzz_qp_box.py, every skill of the sweep"; grep -E "^ *[0-9]+ (ur5|iiwa)" $out/fuzz_qp_box_16_209.log | cut -c1-200
} > $s
cat $s | head -40
