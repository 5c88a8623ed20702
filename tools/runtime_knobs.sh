#!/bin/bash
# Round 5: what the HIP runtime's own settings do to the dependent-launch boundary (1.1 - 1.3 us of every 3 - 4 us tick,
# profiles/r4_team_variants.md).  The headline tick (config 3, 16384 instances, hipGraph of 2000 ticks) and the config-2
# tick under each setting; every run is a fresh process (the runtime reads its flags once).
#   tools/runtime_knobs.sh > gpurun_out/r5_runtime_knobs.txt
cd "$(dirname "$0")/.."
run() {
    label="$1"; shift
    for wl in stack pose; do
        line=$(env "$@" timeout 90 python bench.py --workload $wl --extras 0 --cpu-baseline 0 --min-timed-ms 600 --ramp-ms 150 2>/dev/null | tail -1)
        us=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.3f us/tick (events %.3f)' % (d['ms_per_step']*1e3, d['roofline']['tick_us']))" "$line" 2>/dev/null || echo "FAILED: ${line:0:200}")
        printf '%-58s %-6s %s\n' "$label" "$wl" "$us"
    done
}
if [ -z "$KNOBS_SKIP_DONE" ]; then
run "defaults" CLIK_NOOP=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
fi
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "ROC_USE_FGS_KERNARG=0" ROC_USE_FGS_KERNARG=0
run "ROC_SKIP_KERNEL_ARG_COPY=1" ROC_SKIP_KERNEL_ARG_COPY=1
run "DEBUG_HIP_GRAPH_BATCH_SIZE=1000" DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run "ROC_ACTIVE_WAIT_TIMEOUT=100000" ROC_ACTIVE_WAIT_TIMEOUT=100000
run "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1
# (ROC_SYSTEM_SCOPE_SIGNAL=0 - release fences at agent scope - never finishes its first bracket: the host waits for a
# completion signal it cannot see; measured round 5, the run was killed after 13 minutes)
