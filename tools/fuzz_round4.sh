#!/bin/bash
# Round-4 randomised parity sweeps on the GPU box, every tool under the ONE stated rule (tests/tolerances.py).
#   bash tools/fuzz_round4.sh [out = gpurun_out/r4fuzz]   ->  <out>/r4_fuzz_summary.txt (+ the full logs beside it)
out=${1:-gpurun_out/r4fuzz}
mkdir -p $out
run() { name=$1; shift; python "$@" > $out/$name.log 2>&1; echo "$name rc $?"; }
run fuzz_parity_40_51   tools/fuzz_parity.py 40 51
run fuzz_parity_60_31   tools/fuzz_parity.py 60 31
run fuzz_team_30_7      tools/fuzz_team.py 30 7
run fuzz_team_20_3      tools/fuzz_team.py 20 3
run fuzz_qp_box_16_9    tools/fuzz_qp_box.py 16 9
run fuzz_qp_mixed_30_13 tools/fuzz_qp_mixed.py 30 13
run fuzz_qp_dynamic_200_5 tools/fuzz_qp_dynamic.py 200 5
run fuzz_qp_wide_40_0 tools/fuzz_qp_wide.py 40 0 96
FUZZ_ANGLES=1 python tools/fuzz_parity.py 40 77 > $out/fuzz_parity_angles_40_77.log 2>&1; echo "fuzz_parity_angles rc $?"
s=$out/r4_fuzz_summary.txt
{
echo "# Round 4 randomised parity sweeps on one MI355X (final kernels; every tool holds every instance to the stated rule"
echo "# err <= max(1e-12, 8 u kappa) of tests/tolerances.py; full logs are scratch under $out)"
for f in fuzz_parity_40_51 fuzz_parity_60_31; do
  echo; echo "## tools/fuzz_parity.py  (log $f)"
  grep -c "MISMATCH" $out/$f.log | sed 's/^/instances beyond the rule (MISMATCH lines): /'
  grep -o "([0-9.]* x tol)" $out/$f.log | tr -d '(' | sort -g | tail -1 | sed 's/^/worst err \/ tol of a pinv skill: /'
  tail -2 $out/$f.log
done
for f in fuzz_team_30_7 fuzz_team_20_3; do
  echo; echo "## tools/fuzz_team.py  (log $f)"
  grep -o "([0-9.]* x tol" $out/$f.log | tr -d '(' | sort -g | tail -1 | sed 's/^/worst err \/ tol: /'
  tail -1 $out/$f.log
done
echo; echo "## tools/fuzz_qp_box.py  (log fuzz_qp_box_16_9)"; tail -1 $out/fuzz_qp_box_16_9.log
echo; echo "## tools/fuzz_qp_mixed.py  (log fuzz_qp_mixed_30_13)"; grep "skipped" $out/fuzz_qp_mixed_30_13.log | cut -c1-200; tail -1 $out/fuzz_qp_mixed_30_13.log
echo; echo "## tools/fuzz_qp_dynamic.py  (log fuzz_qp_dynamic_200_5)"; tail -1 $out/fuzz_qp_dynamic_200_5.log
echo; echo "## tools/fuzz_qp_wide.py  (log fuzz_qp_wide_40_0)"; tail -1 $out/fuzz_qp_wide_40_0.log
echo; echo "## FUZZ_ANGLES=1 tools/fuzz_parity.py 40 77  (generated constraints also draw atan2 / asin / acos / atan / tanh / fmin / fmax)"
echo "instances beyond the rule (MISMATCH lines): $(grep -c MISMATCH $out/fuzz_parity_angles_40_77.log); skills whose constraints ran as generated device code: $(grep -c 'pinv dynamic refused: the skill has constraint expressions' $out/fuzz_parity_angles_40_77.log) of 40"; tail -2 $out/fuzz_parity_angles_40_77.log
echo; echo "## tools/fuzz_qp_mixed.py, every skill of the sweep"; grep -E "^ *[0-9]+ (ur5|iiwa)" $out/fuzz_qp_mixed_30_13.log | cut -c1-200
echo; echo "## tools/fuzz_qp_box.py, every skill of the sweep"; grep -E "^ *[0-9]+ (ur5|iiwa)" $out/fuzz_qp_box_16_9.log | cut -c1-200
} > $s
cat $s | head -40
