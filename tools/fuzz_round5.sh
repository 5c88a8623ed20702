#!/bin/bash
# Round-5 randomised parity sweeps on the GPU box, every tool under the ONE stated rule (tests/tolerances.py).
#   bash tools/fuzz_round5.sh [out = gpurun_out/r5fuzz]   ->  <out>/r5_fuzz_summary.txt (+ the full logs beside it)
out=${1:-gpurun_out/r5fuzz}
mkdir -p $out
run() { name=$1; shift; python "$@" > $out/$name.log 2>&1; echo "$name rc $?"; }
run fuzz_parity_40_151   tools/fuzz_parity.py 40 151
run fuzz_parity_60_131   tools/fuzz_parity.py 60 131
run fuzz_team_30_107      tools/fuzz_team.py 30 107
run fuzz_team_20_103      tools/fuzz_team.py 20 103
run fuzz_qp_box_16_109    tools/fuzz_qp_box.py 16 109
run fuzz_qp_mixed_30_113 tools/fuzz_qp_mixed.py 30 113
run fuzz_qp_dynamic_200_105 tools/fuzz_qp_dynamic.py 200 105
run fuzz_qp_wide_40_100 tools/fuzz_qp_wide.py 40 100 96
FUZZ_ANGLES=1 python tools/fuzz_parity.py 40 177 > $out/fuzz_parity_angles_40_177.log 2>&1; echo "fuzz_parity_angles rc $?"
s=$out/r5_fuzz_summary.txt
{
echo "# Round 5 randomised parity sweeps on one MI355X (final kernels; every tool holds every instance to the stated rule"
echo "# err <= max(1e-12, 8 u kappa) of tests/tolerances.py; full logs are scratch under $out)"
for f in fuzz_parity_40_151 fuzz_parity_60_131; do
  echo; echo "## tools/fuzz_parity.py  (log $f)"
  grep -c "MISMATCH" $out/$f.log | sed 's/^/instances beyond the rule (MISMATCH lines): /'
  grep -o "([0-9.]* x tol)" $out/$f.log | tr -d '(' | sort -g | tail -1 | sed 's/^/worst err \/ tol of a pinv skill: /'
  tail -2 $out/$f.log
done
for f in fuzz_team_30_107 fuzz_team_20_103; do
  echo; echo "## tools/fuzz_team.py  (log $f)"
  grep -o "([0-9.]* x tol" $out/$f.log | tr -d '(' | sort -g | tail -1 | sed 's/^/worst err \/ tol: /'
  tail -1 $out/$f.log
done
echo; echo "## tools/fuzz_qp_box.py  (log fuzz_qp_box_16_109)"; tail -1 $out/fuzz_qp_box_16_109.log
echo; echo "## tools/fuzz_qp_mixed.py  (log fuzz_qp_mixed_30_113)"; grep "skipped" $out/fuzz_qp_mixed_30_113.log | cut -c1-200; tail -1 $out/fuzz_qp_mixed_30_113.log
echo; echo "## tools/fuzz_qp_dynamic.py  (log fuzz_qp_dynamic_200_105)"; tail -1 $out/fuzz_qp_dynamic_200_105.log
echo; echo "## tools/fuzz_qp_wide.py  (log fuzz_qp_wide_40_100)"; tail -1 $out/fuzz_qp_wide_40_100.log
echo; echo "## FUZZ_ANGLES=1 tools/fuzz_parity.py 40 177  (generated constraints also draw atan2 / asin / acos / atan / tanh / fmin / fmax)"
echo "instances beyond the rule (MISMATCH lines): $(grep -c MISMATCH $out/fuzz_parity_angles_40_177.log); skills whose constraints ran as generated device code: $(grep -c 'pinv dynamic refused: the skill has constraint expressions' $out/fuzz_parity_angles_40_177.log) of 40"; tail -2 $out/fuzz_parity_angles_40_177.log
echo; echo "## tools/fuzz_qp_mixed.py, every skill of the sweep"; grep -E "^ *[0-9]+ (ur5|iiwa)" $out/fuzz_qp_mixed_30_113.log | cut -c1-200
echo; echo "## tools/fuzz_qp_box.py, every skill of the sweep"; grep -E "^ *[0-9]+ (ur5|iiwa)" $out/fuzz_qp_box_16_109.log | cut -c1-200
} > $s
cat $s | head -40
