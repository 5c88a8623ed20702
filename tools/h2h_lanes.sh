mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_team.py -x -q > gpurun_out/r2b/pytest_team.log 2>&1; tail -15 gpurun_out/r2b/pytest_team.log
for B in 4096 16384 32768 65536; do for L in 1 4; do
python bench.py --batch $B --lanes $L --cpu-baseline 0 --steps 1000 --warmup 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('B=%d lanes=%s kernel=%s us/tick=%.3f kernel_us=%.3f value=%.4g' % ($B,'$L',d['config']['kernel'],d['ms_per_step']*1e3,d['roofline']['kernel_us'],d['value']))" | tee -a gpurun_out/r2b/h2h.txt
done; done
