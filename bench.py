#!/usr/bin/env python3
"""Headline benchmark: CLIK steps/sec on the 7-DoF 3-task priority stack.

    python bench.py --gpus N --steps K --warmup W

A *step* is one controller tick over the whole batch resident in HBM (one
kernel launch per tick per GPU).  `value` = instance-steps per second =
(N * batch * steps timed) / time, the BASELINE.json metric "CLIK steps/sec (whole
node), 7-DoF 3-task priority stack, batch 16384" (batch is per GPU: weak
scaling, as in BASELINE config 5 = 131072 instances over 8 GPUs;
`--global-batch G` cuts ONE batch of G instances over the ranks instead: strong
scaling, reported as such).

The JSON line also carries
  roofline      algorithmic HBM bytes (q 56 B + target 56 B +
                dq 56 B + mode 4 B per instance-step; the QP: + 48 B slack + 4 B
                status) over the per-tick time from HIP events on the launch
                stream, vs 8 TB/s; `tick_us` is that per-tick time (kernel body
                + the dependent-launch boundary: an upper bound of the kernel's
                duration); `kernel_body_us` is the kernel's own first-wave-start
                to last-wave-end time where profiles/ holds a stamp measurement
                of this configuration (with its source); `fp64` is the executed
                fp64 VALU rate from the PMC passes under profiles/ against the
                78.6 TF vector peak, and `binds` names the bound that applies
  cpu_baseline  the C restatement of the reference algorithm (oracle/, kind
                "port") timed on this host's cores on a bounded sample
  extras        (default N=1 run only) the other BASELINE configurations under
                the same clock: config 2 (pose, 4096 and 16384), config 4 (QP,
                16384: cold, and hot-started as the reference's solver is after
                its first tick), config 3 at the config-5 batch (131072), each with
                ms_per_step / kernel / roofline / cpu_baseline; the on-device
                rollouts of both controllers (256 / 64 ticks per launch); and
                config 3 at 16384 as resident ticks fed ahead (one launch,
                device-side tickets)

Timing protocol: W untimed warm-up steps, then an untimed, time-based clock ramp
(--ramp-ms of replays: a fresh GPU needs ~100 ms of work before its clocks
settle, which a 20-step run never reaches), then R back-to-back replays of a
hipGraph inside ONE barrier + synchronize bracket.  The graph holds the K steps
repeated to >= 1024 ticks (a graph launch opens with a ~10 us device bubble that
a 20-tick graph would pay every 20 ticks).  `ms_per_step` = bracket wall time /
timed steps (max over ranks), `value` = instances * timed steps / wall;
`roofline.tick_us` = median over the R replays of (HIP-event time of one
replay / its ticks).  R is chosen so that the bracket holds >= --min-timed-ms of
work (default 2 s for the headline) and >= 50 replays; R and the timed steps are
stated in `config`.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (torch.distributed.run on 127.0.0.1) BEFORE anything touches a GPU and
relays rank 0's JSON line; with fewer than N visible devices it exits non-zero.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TF = 78.6       # half of the 157.3 TF fp32 vector peak
PROFILE_FILES = ("r6_counters.json", "r5_counters.json", "r4_counters.json", "r3_counters.json", "r2_counters.json",
                 "r1_traffic.json")     # newest first
BODY_FILES = ("r6_body_time.json", "r4_body_time_light.json", "r3_body_time.json")
ISSUE_PROBE_FILES = ("r6_fp64_issue_ceiling.json", "r5_fp64_issue_ceiling.json")
METRIC = "CLIK steps/sec (whole node), 7-DoF 3-task priority stack, batch 16384"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=16384, help="instances per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="G > 0: ONE batch of G instances cut into contiguous shards over the ranks "
                         "(strong scaling; BASELINE config 5 = 131072); overrides --batch")
    ap.add_argument("--workload", default="stack", choices=["stack", "pose", "qp"])
    ap.add_argument("--dist", default="mixed", choices=["interior", "mixed"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", type=int, default=1,
                    help="replay the K ticks from one hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--allgather", type=int, default=0,
                    help="also time the RCCL all-gather of dq (its own bracket, and tick + all-gather; "
                         "reported separately, never folded into `value`)")
    ap.add_argument("--ticks-per-launch", type=int, default=1,
                    help="K > 1: the on-device rollout (K ticks of solve -> clamp -> Euler per launch, "
                         "SURVEY.md 8(d) 'launch-amortised'); steps must be a multiple of K")
    ap.add_argument("--qp-hot", type=int, default=0,
                    help="qp workload: carry each instance's working set from tick to tick (hot start); 1 = the target moves "
                         "1 mm per tick, 2 = the target stands still (no working-set change: the best case)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="lanes per robot instance of the pinv kernel: 0 = the library's choice, 1 = "
                         "lane-per-instance kernels only, 4 / 8 / 16 = the multi-lane kernel (CLIK_LANES)")
    ap.add_argument("--ramp-ms", type=float, default=250.0, help="untimed clock ramp before the timed region")
    ap.add_argument("--min-timed-ms", type=float, default=2000.0, help="least work inside the timed bracket")
    ap.add_argument("--replays", type=int, default=0, help="R (0: from --min-timed-ms, at least 50)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=3.0,
                    help="bound of the cpu_baseline leg of the headline (each extras entry: a third of it, >= 1 s)")
    ap.add_argument("--ring", type=int, default=4,
                    help="input / output buffers the ticks of a graph rotate through (a different synthetic batch in "
                         "each; 1 = every tick re-reads the same rows)")
    ap.add_argument("--extras", type=int, default=-1,
                    help="1: append the other BASELINE configurations (see the module text) as `extras`; "
                         "-1 = only for the default headline invocation on one GPU; 0 = never")
    ap.add_argument("--extras-timed-ms", type=float, default=400.0)
    ap.add_argument("--full", type=int, default=0,
                    help="1: print the long form of the line (every timing detail and source text) instead of the "
                         "compact one; the compact line names where each number comes from by file")
    ap.add_argument("--strong-batch", type=int, default=131072,
                    help="N > 1: also time ONE batch of this many instances cut over the ranks (BASELINE config 5; 0 = skip)")
    ap.add_argument("--big-batch", type=int, default=131072,
                    help="N > 1: also time this many instances PER GPU (the regime where a GPU is busy; 0 = skip)")
    return ap.parse_args()


_T0 = time.time()


def phase(text):
    """wall-clock stamps of the run's phases on stderr (to align a driver's GPU-utilisation samples with them)"""
    sys.stderr.write("[bench %8.2f s] %s\n" % (time.time() - _T0, text))
    sys.stderr.flush()


def self_launch(args):
    """--gpus N without a launcher: become the launcher.  Nothing here initialises a GPU
    (torch.cuda.device_count() only counts), and the ranks are fresh child processes."""
    import torch
    n = torch.cuda.device_count()
    if os.environ.get("CLIK_BENCH_SHARED_GPU", "0") == "1":
        n = max(n, args.gpus) if n >= 1 else n
    if n < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; not running fewer ranks "
                         "under that label\n" % (args.gpus, n))
        sys.exit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def make_workload(name, fk):
    import casclik_amd as cc
    from casclik_amd import skills
    if name == "stack":
        spec, opts = skills.stack_skill(fk), dict(skills.STACK_OPTIONS)
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
    elif name == "pose":
        spec, opts = skills.pose_skill(fk), None
        ctrl = cc.PseudoInverseController(skill_spec=spec)
    else:
        spec, opts = skills.qp_skill(fk), None
        ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    return spec, opts, ctrl


def host_cores():
    """(cores this process may be scheduled on, CPU-time quota of its cgroup in cores or None)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    for fn in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(fn) as f:
                parts = f.read().split()
            if fn.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = float(parts[0]) / float(parts[1])
            else:
                q = float(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        quota = q / float(f.read().split()[0])
            break
        except Exception:
            continue
    return avail, quota


_CPU_CACHE = {}


def cpu_baseline(workload, spec, opts, Q, Y, seconds):
    """Time the C restatement of the reference algorithm (oracle/clik_oracle_c.c, OpenMP over
    instances) on the host cores this process may use: single core and the best thread count, each
    the median of 3 repeats with the spread stated.  (The bound of the sweep: a cgroup CPU quota caps
    what more threads can buy; it is reported when there is one.)"""
    import numpy as np
    if workload in _CPU_CACHE:
        return dict(_CPU_CACHE[workload], note="same skill and sample as the entry above")
    avail, quota = host_cores()
    from oracle import c_oracle
    # (descriptors written down in oracle/baseline_desc.py, not lowered by the product's front-end)
    if workload == "qp":
        co = c_oracle.CQpOracle(None, baseline=("iiwa", "qp"))
        rows = 16384
        what = ("C restatement of reactive_qp.py:175-246 + dense Goldfarb-Idnani on the full QP "
                "(oracle/clik_oracle_c.c::orc_qp_solve_batch; the reference hands the same H, A, lbA, ubA to qpOASES)")
    else:
        co = c_oracle.CPinvOracle(None, opts, baseline=("iiwa", workload))
        rows = 8 * 16384          # enough rows per thread to amortise the OpenMP fork/join
        what = "C restatement of the reference algorithm (oracle/clik_oracle_c.c)"
    reps_rows = max(1, rows // len(Q))
    Qs, Ys = np.tile(Q, (reps_rows, 1))[:rows], np.tile(Y, (reps_rows, 1))[:rows]
    sample = len(Qs)

    def rate(threads, n_rows):
        t0 = time.perf_counter()
        co.solve_batch(0.0, Qs[:n_rows], Y=Ys[:n_rows], nthreads=threads)
        return n_rows / (time.perf_counter() - t0)

    budget_t0 = time.perf_counter()
    one_rows = min(sample, 4096)
    rate(1, 256)                                   # (page the library in)
    single = sorted(rate(1, one_rows) for _ in range(3))
    # Thread counts to try.  A cgroup CPU quota (this pool's GPU boxes: 256 schedulable cores, quota 16) makes a
    # short burst on many threads look fast and then throttles the group: every candidate is therefore timed
    # SUSTAINED (back-to-back ticks for its share of the budget, median), and candidates far beyond the quota
    # are not tried.
    cand = {min(8, avail), min(16, avail), min(32, avail), min(64, avail), min(128, avail), avail}
    if quota:
        q = max(1, int(round(quota)))
        cand = {c for c in cand if c <= 2 * q} | {min(q, avail), min(2 * q, avail)}
    cand = sorted(cand)
    share = max(0.5, 0.8 * (seconds - (time.perf_counter() - budget_t0)) / len(cand))
    sweep, runs = {}, {}
    for threads in cand:
        rate(threads, sample)                      # first touch / thread start-up
        t_c, vals = time.perf_counter(), []
        while len(vals) < 3 or time.perf_counter() - t_c < share:
            vals.append(rate(threads, sample))
        vals.sort()
        sweep[threads], runs[threads] = vals[len(vals) // 2], vals
    best_threads = max(sweep, key=sweep.get)
    reps = runs[best_threads]
    n_ticks = len(reps)
    out = {"value": sweep[best_threads], "unit": "instance-steps/s", "cores": best_threads, "kind": "port",
           "host_cores_available": avail, "cgroup_cpu_quota_cores": quota,
           "repeats": n_ticks, "spread": [reps[0], reps[-1]],
           "single_core": {"value": single[1], "spread": [single[0], single[-1]], "rows": one_rows},
           "thread_sweep_sustained": {str(k): v for k, v in sorted(sweep.items())},
           "sample": "%d back-to-back ticks of a %d-instance batch (the bench batch tiled %dx) at the best thread count "
                     "of a sustained sweep, median; %s, OpenMP over instances" % (n_ticks, sample, reps_rows, what)}
    _CPU_CACHE[workload] = out
    return out


def profiled(workload, dist_name, batch, kernel, hot=False):
    """Per-launch PMC figures of this configuration recorded under profiles/ (they cannot be collected
    from inside this process: separate rocprofv3 --pmc passes, tools/rocprof_counters.py).  Returns
    (entry, source) or (None, None): a configuration that was not profiled gets no traffic figure."""
    keys = ["%s_%s_B%d_%s" % (workload, dist_name, batch, kernel)]
    if hot:
        # a hot-started tick runs fewer passes than a cold one: only its OWN counter passes count (none: null)
        keys = [keys[0] + "_hot"]
    elif str(kernel).endswith(("/lane", "/mp2")) or "/" not in str(kernel):
        keys.append("%s_%s_B%d" % (workload, dist_name, batch))       # round-1 entries (recorded before variants had names)
    for fn in PROFILE_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                prof = json.load(f)
        except Exception:
            continue
        for k in keys:
            ent = prof.get(k)
            if isinstance(ent, dict) and ("traffic_bytes" in ent or "fp64_flops_per_launch" in ent):
                want = ent.get("kernel")
                if want is not None and want != kernel:
                    continue
                return ent, "profiles/" + fn + "#" + k
    return None, None


def body_time(workload, dist_name, batch, kernel, hot=False):
    """Kernel body time (first wave start to last wave end, s_memrealtime stamps of the CLIK_STAMP build,
    tools/stamp_body.py) of this configuration if profiles/ holds one."""
    key = "%s_%s_B%d_%s%s" % (workload, dist_name, batch, kernel, "_hot" if hot else "")
    for fn in BODY_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                ent = json.load(f).get(key)
        except Exception:
            continue
        if isinstance(ent, dict) and "body_us_median" in ent:
            return ent, "profiles/" + fn + "#" + key
    return None, None


WORKLOAD_TEXT = {
    "stack": "BASELINE config 3: %d x KUKA iiwa 7-DoF per GPU, priority stack [multidim joint-limit set; 6-D pose; "
             "joint centering], PseudoInverseController",
    "pose": "BASELINE config 2: %d x iiwa, single 6-D pose task, PseudoInverseController",
    "qp": "BASELINE config 4: %d x iiwa ReactiveQPController (soft 6-D pose + joint-speed VelocitySetConstraint)",
}


class Ctx(object):
    def __init__(self, rank, world, dev, dist, shared_gpu=False):
        self.rank, self.world, self.dev, self.dist = rank, world, dev, dist
        # CLIK_BENCH_SHARED_GPU=1 (a smoke test of the multi-rank code path on a one-GPU box): every rank on
        # cuda:0, process group "gloo", control-plane reductions and the all-gather on host tensors
        self.shared_gpu = shared_gpu

    def reduce_max(self, values, as_int=False):
        """element-wise MAX over ranks of a short list of numbers"""
        import torch
        if self.dist is None:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.int64 if as_int else torch.float64,
                         device="cpu" if self.shared_gpu else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [int(v) if as_int else float(v) for v in t.tolist()]


def time_allgather(ctx, tick, dQ, n_total, M=200):
    """the optional all-gather of dq: its own bracket, then tick + all-gather back to back on the same
    stream (eager launches: a collective is not captured); reported apart, never folded into `value`"""
    import torch
    from casclik_amd.distributed import all_gather_rows as _agr
    dist = ctx.dist
    if dist is None:
        return {"error": "no process group"}

    def all_gather_rows(t, n):          # (gloo: host copies; RCCL: the device rows)
        return _agr(t.cpu() if ctx.shared_gpu else t, n)
    full = all_gather_rows(dQ, n_total)
    for _ in range(20):
        all_gather_rows(dQ, n_total)
    dist.barrier()
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(M):
        all_gather_rows(dQ, n_total)
    torch.cuda.synchronize()
    dist.barrier()
    ag_us = (time.perf_counter() - ta) * 1e6 / M
    for _ in range(20):
        tick()
        all_gather_rows(dQ, n_total)
    dist.barrier()
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for _ in range(M):
        tick()
        all_gather_rows(dQ, n_total)
    torch.cuda.synchronize()
    dist.barrier()
    both_us = (time.perf_counter() - ta) * 1e6 / M
    tt = ctx.reduce_max([ag_us, both_us])
    return {"allgather_us": round(float(tt[0]), 2), "tick_plus_allgather_us_eager": round(float(tt[1]), 2),
            "bytes_per_rank": int(dQ.numel() * 8), "bytes_gathered": int(full.numel() * 8),
            "backend": "gloo (host copies)" if ctx.shared_gpu else "nccl (RCCL all_gather_into_tensor over xGMI)", "calls": M}


DRAWN_MAX = 131072
CHECK_ROWS = 192


def check_outputs(workload, spec, opts, Q, Y, out_rows, slack_rows=None):
    """One output slot of the timed region against the numpy oracle on its first CHECK_ROWS instances (the
    bracket itself must be seen to have produced the reference's numbers, not only smoke()'s separate call).  The oracle is
    the checker here, never the thing measured.  Returns {"ok", "instances", "rule", "worst_err_over_tol"}."""
    import numpy as np
    from oracle import clik_oracle
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    import tolerances
    n = min(CHECK_ROWS, len(Q))
    got = np.asarray(out_rows[:n], dtype=float)
    if workload == "qp":
        ref, _, rslack, rstatus = clik_oracle.qp_solve_batch(spec, 0.0, Q[:n], Y=Y[:n])
        ok = tolerances.qp_close(got, ref, rows=rstatus == 0)
    else:
        ref, _ = clik_oracle.pinv_solve_batch(spec, opts or {}, 0.0, Q[:n], Y=Y[:n])
        ok = tolerances.pinv_close(got, ref)
    last = dict(tolerances.LAST)
    return {"ok": bool(ok), "instances": int(last.get("checked", n)), "rule": last.get("rule"),
            "worst_err_over_tol": None if last.get("worst") is None else round(float(last["worst"]), 4),
            "what": "ring slot 0's output after the timed bracket vs oracle/clik_oracle.py (tests/tolerances.py rule)"}


def measure(ctx, fk, workload, B, dist_name, seed, K, W, TPL=1, qp_hot=0, use_graph=1, ramp_ms=250.0,
            min_timed_ms=2000.0, replays=0, allgather=0, global_batch=0, ring=4):
    """One timed configuration.  Returns (entry dict, (spec, opts, Q, Y)) on every rank; the entry is
    complete on rank 0."""
    import torch
    from casclik_amd import skills
    from casclik_amd.distributed import shard_bounds
    dev, dist, world, rank = ctx.dev, ctx.dist, ctx.world, ctx.rank
    spec, opts, ctrl = make_workload(workload, fk)
    if global_batch:
        # ONE batch, contiguous shards (casclik_amd.distributed.shard_bounds): every rank draws the same
        # global inputs and keeps its rows
        Qg, Yg = skills.synthetic_inputs(fk, global_batch, seed=seed, distribution=dist_name)
        lo, hi = shard_bounds(global_batch, rank, world)
        Q, Y = Qg[lo:hi], Yg[lo:hi]
        B = hi - lo
    else:
        # every rank owns its own shard of the (weak-scaled) global batch
        # (beyond 131072 instances: that many drawn, tiled - the host-side draw evaluates a forward kinematics per row
        # in Python, 65 us each)
        drawn = min(B, DRAWN_MAX)
        Q, Y = skills.synthetic_inputs(fk, drawn, seed=seed + 1000 * rank, distribution=dist_name)
        if drawn < B:
            import numpy as np
            reps = (B + drawn - 1) // drawn
            Q, Y = np.tile(Q, (reps, 1))[:B].copy(), np.tile(Y, (reps, 1))[:B].copy()
    Qd = torch.from_numpy(Q).to(dev)
    Yd = torch.from_numpy(Y).to(dev)
    dQ = torch.empty((B, Q.shape[1]), dtype=torch.float64, device=dev)
    # successive ticks of a graph rotate through `ring` input / output buffers (slot 0 = the batch drawn above, slot s =
    # the same rows rolled by s * B / ring instances: another instance on every lane, no second host-side draw), so that
    # no tick finds the lines the tick before it touched; the QP's hot start keeps one working set per slot
    ring = max(1, int(ring)) if (TPL == 1 and B > 1) else 1
    slots = [(Qd, Yd, dQ)]
    moving_target = bool(qp_hot) and workload == "qp"
    settled = moving_target and int(qp_hot) == 2     # (the target stands still: the working set handed over is the optimal one)
    for s_ in range(1, ring):
        if moving_target:
            # hot-started ticks: the SAME instances in every slot (distinct memory), the target moved by a millimetre per slot
            # along x, y, z in turn, ONE working set handed from tick to tick - what a control loop following a moving target
            # sees: working sets that are close to, not always equal to, the next tick's (each slot keeping
            # its own set re-solved a QP whose optimal set it already held - zero pivots, a best case)
            y_s = Yd.clone()
            if not settled:
                y_s[:, (s_ - 1) % 3] += 1e-3 * s_
            slots.append((Qd.clone(), y_s.contiguous(), torch.empty_like(dQ)))
            continue
        shift = (s_ * B) // ring
        slots.append((torch.roll(Qd, shift, 0).contiguous(), torch.roll(Yd, shift, 0).contiguous(), torch.empty_like(dQ)))

    if TPL > 1:
        if K % TPL or W % TPL:
            raise SystemExit("--ticks-per-launch needs steps and warmup divisible by it")
        times = [0.0] * TPL
        state = {"q": Qd}

        def tick():         # one launch = TPL ticks; the state is carried from launch to launch
            state["q"] = ctrl.rollout_batch(times, state["q"], input_var=Yd, dt=1e-3, max_speed=2.0)[0]
        use_graph = 0
        K //= TPL
        W //= TPL
    else:
        kw = dict(hot_start=bool(qp_hot)) if workload == "qp" else {}
        if moving_target:
            kw["hot_set"] = torch.zeros((B,), dtype=torch.int32, device=dev)
        bound = [ctrl.bind_batch(q_, input_var=y_, out=o_, **kw) for q_, y_, o_ in slots]
        if moving_target:
            bound[0]()                      # (a cold tick fills the shared working sets; from here on every tick is hot)
            for b_ in bound:
                b_.primed = True
        turn = {"i": 0}

        def tick():
            bound[turn["i"] % len(bound)]()
            turn["i"] += 1

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for _ in range(W):
            tick()
        stream.synchronize()
        graph = None
        # a replay = GK steps = the K steps repeated until the graph holds >= 1024 ticks: every graph launch
        # starts with a ~10 us bubble on the device (measured: 6.26 us per step from a 20-tick graph against
        # 5.74 us from a 2000-tick graph of the same kernel), which a control loop that enqueues ticks
        # continuously never sees
        GK = K * max(1, -(-1024 // K)) if (use_graph and TPL == 1) else K
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(GK):
                    tick()
            graph.replay()          # untimed first replay (upload)
            stream.synchronize()

        def replay():               # GK steps
            if graph is not None:
                graph.replay()
            else:
                for _ in range(GK):
                    tick()

        # untimed clock ramp; its rate also sizes R
        t_r = time.perf_counter()
        n_ramp = 0
        while True:
            replay()
            n_ramp += 1
            if n_ramp % 4 == 0 or GK >= 500:
                stream.synchronize()
                if (time.perf_counter() - t_r) * 1e3 >= ramp_ms:
                    break
        stream.synchronize()
        est_ms = (time.perf_counter() - t_r) * 1e3 / n_ramp          # one replay, host-inclusive (upper bound)
        R = replays if replays > 0 else max(50, int(min_timed_ms / max(est_ms, 1e-3)) + 1)
        R = min(R, 200000)
        R = ctx.reduce_max([R], as_int=True)[0]
        n_ev = min(R, 2000)                                           # events around the first n_ev replays
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
        ev_end = torch.cuda.Event(enable_timing=True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record(stream)
        for r in range(R):
            replay()
            if r < n_ev:
                evs[r + 1].record(stream)
        ev_end.record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0

        # the optional all-gather of dq: its own bracket, then tick + all-gather
        # back to back on the same stream (eager launches: a collective is not captured)
        ag = None
        if allgather and world > 1:
            # (a collective that fails must not cost the compute line: the error is carried in its place, and the
            # ranks re-align on the host-side barrier of reduce_max below only if the group still works)
            try:
                ag = time_allgather(ctx, tick, dQ, global_batch if global_batch else world * B)
            except Exception as exc:
                ag = {"error": repr(exc)[:300], "backend": "gloo (host copies)" if ctx.shared_gpu else "nccl (RCCL)"}

    per_replay_ms = sorted(evs[r].elapsed_time(evs[r + 1]) for r in range(n_ev))
    med_ms = per_replay_ms[n_ev // 2]
    dev_ms = evs[0].elapsed_time(ev_end)
    wall, dev_ms, med_ms = ctx.reduce_max([wall, dev_ms, med_ms])

    K, W, GK = K * TPL, W * TPL, GK * TPL          # report in ticks
    timed_steps = R * GK
    n_inst_total = global_batch if global_batch else world * B
    value = n_inst_total * timed_steps / wall
    tick_us = med_ms * 1e3 / GK
    n_q, n_y = Q.shape[1], Y.shape[1]
    if workload == "qp":
        n_slack = int(getattr(spec, "n_slack_var", 0) or 0)
        bytes_per_inst = 8 * (n_q + n_y + n_q) + 8 * n_slack + 4          # q, target, dq, slack, status
    else:
        bytes_per_inst = 8 * (n_q + n_y + n_q) + 4                        # q, target, dq, mode
    alg_bytes = bytes_per_inst * B
    kernel = getattr(ctrl, "kernel_name", None)
    variant = getattr(ctrl, "kernel_variant", None)
    if callable(variant):
        kernel = variant(B, hot=True) if ((qp_hot or TPL > 1) and workload == "qp") else variant(B)
    if TPL == 1:
        achieved = alg_bytes / (tick_us * 1e-6) / 1e9
        prof, prof_src = profiled(workload, dist_name, B, kernel, hot=bool(qp_hot))
        body, body_src = body_time(workload, dist_name, B, kernel, hot=bool(qp_hot))
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": prof.get("traffic_bytes") if prof else None,
                "traffic_source": prof_src if (prof and "traffic_bytes" in prof) else None,
                "tick_us": tick_us, "tick_us_mean": dev_ms * 1e3 / timed_steps,
                "tick_us_is": "HIP events on the launch stream around each graph replay / its ticks: kernel body + "
                              "the dependent-launch boundary (an upper bound of the kernel's duration)",
                # first wave start -> last wave end (s_memrealtime stamps, tools/stamp_body.py): the stamped build's
                # own body, the dependent-launch boundary it leaves of its tick, and this run's tick minus that
                # boundary = the shipped kernel's body
                "kernel_body_us": (tick_us - body["boundary_us"]) if body else None,
                "kernel_body_us_stamped_build": body.get("body_us_median") if body else None,
                "launch_boundary_us": body.get("boundary_us") if body else None,
                "kernel_body_source": body_src,
                "algorithmic_bytes_per_instance": bytes_per_inst,
                "algorithmic_bytes_per_launch": alg_bytes,
                # which numbers this process MEASURED and which it read from a file a profiler run left under profiles/
                "provenance": {"tick_us / achieved / frac": "measured (this run: HIP events on the launch stream)",
                               "traffic": ("recorded, " + prof_src) if (prof and "traffic_bytes" in prof) else None,
                               "fp64_frac": ("recorded flops per launch (" + prof_src + ") / measured tick") if prof else None,
                               "kernel_body_us": ("measured tick - recorded launch boundary (" + body_src + ")") if body else None,
                               "issue_probe_body_us": "recorded (profiles/*_fp64_issue_ceiling.json)"}}
        # what the ISSUE of the kernel's fp64 instructions costs on this machine with nothing else going on (bare-FMA waves
        # in the same launch shape and register allocation, tools/probe_fp64_peak.hip; recorded, not measured here)
        probe, probe_fn = None, None
        for probe_fn in ISSUE_PROBE_FILES:
            try:
                with open(os.path.join(ROOT, "profiles", probe_fn)) as f:
                    probe = json.load(f).get("%s_%s_B%d_%s" % (workload, dist_name, B, kernel))
            except Exception:
                probe = None
            if probe:
                break
        if probe:
            roof["issue_probe"] = dict(probe, source="profiles/" + probe_fn)
        if prof and "fp64_flops_per_launch" in prof:
            # executed fp64 flops (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 x 64 lanes, FMA = 2) over the measured
            # tick time: what the VALUs did, not what the literal algorithm would need
            tf = prof["fp64_flops_per_launch"] / (tick_us * 1e-6) / 1e12
            roof["fp64"] = {"achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s (executed, vector fp64)",
                            "frac": tf / FP64_VALU_PEAK_TF, "flops_per_launch": prof["fp64_flops_per_launch"],
                            "source": prof_src}
            if "valu_issue_frac" in prof:
                roof["fp64"]["valu_issue_frac"] = prof["valu_issue_frac"]
            roof["binds"] = ("fp64 VALU issue" if roof["fp64"]["frac"] >= roof["frac"] else "hbm")
        else:
            roof["fp64"] = None
            roof["binds"] = None
    else:
        # an on-device rollout reads q / y and writes q once per LAUNCH of TPL ticks: a per-tick HBM
        # fraction would not be a bandwidth
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "tick_us": tick_us, "tick_us_mean": dev_ms * 1e3 / timed_steps,
                "algorithmic_bytes_per_launch": alg_bytes, "ticks_per_launch": TPL,
                "note": "rollout: state stays in registers between ticks; bytes move once per launch"}
    text = WORKLOAD_TEXT[workload] % B
    if workload == "qp" and qp_hot:
        text += " (hot-started from the previous tick's working set; %s)" % (
            "target standing still" if int(qp_hot) == 2 else "target moving")
    entry = {
        "value": value, "unit": "instance-steps/s", "ms_per_step": wall * 1e3 / timed_steps,
        "config": {
            "workload": text,
            "batch_per_gpu": B, "inputs": "%s seed %d%s%s" % (dist_name, seed, "" if B <= DRAWN_MAX or global_batch else
                                                             " (%d drawn, tiled)" % DRAWN_MAX,
                                                             (", ring of %d buffers" % len(slots) + (
                                                                 ": the same instances, the target %s, one working "
                                                                 "set handed from tick to tick" % ("standing still (best case: "
                                                                 "no working-set change)" if settled else "moved 1 mm per slot")
                                                                 if moving_target else ""))
                                                             if len(slots) > 1 else ""),
            "kernel": kernel,
            "launch": ("hipGraph of %d ticks (the K ticks x %d)" % (GK, GK // K) if graph is not None
                       else "eager, one launch per tick") if TPL == 1
                      else "on-device rollout, %d ticks per launch (solve -> clamp -> Euler)" % TPL,
            "replays": R, "timed_steps": timed_steps, "timed_ms": wall * 1e3, "clock_ramp_ms": ramp_ms,
            "timing": "R back-to-back replays in one barrier+synchronize bracket; "
                      "ms_per_step = wall / timed_steps, max over ranks",
            "ticks_per_s": timed_steps / wall,
            "parallelism": "dp%d (independent shards, no data-path collective)" % world,
        },
        "roofline": roof,
    }
    if global_batch:
        entry["config"]["global_batch"] = global_batch
    if rank == 0 and TPL == 1:
        # the bracket's own output (slot 0 holds the answers of the last tick that ran on the drawn batch)
        try:
            torch.cuda.synchronize()
            entry["check"] = check_outputs(workload, spec, opts, Q, Y, slots[0][2].detach().cpu().numpy())
        except Exception as exc:
            entry["check"] = {"ok": False, "error": repr(exc)[:200]}
    if ag is not None:
        entry["allgather"] = ag
    del graph
    return entry, (spec, opts, Q, Y)


def measure_resident(fk, dist_name, seed, B=16384, short=20000, ring=4, integrate=False, workload="stack"):
    """BASELINE config 3 at 16384 instances through clik_pinv_resident_run over a ring of `ring` input / output slots
    (a different synthetic batch in every slot; tick k uses slot (k - 1) % ring) with every ticket published ahead
    (tools/resident_probe.py is the long form, closed loop included)."""
    import torch
    import casclik_amd as cc
    from casclik_amd import skills
    if workload == "pose":
        ctrl = cc.PseudoInverseController(skill_spec=skills.pose_skill(fk))
    else:
        ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    slots = [skills.synthetic_inputs(fk, B, seed=seed + 17 * s, distribution=dist_name) for s in range(ring)]
    Qr = torch.stack([torch.from_numpy(q).cuda() for q, _ in slots]).contiguous()
    Yr = torch.stack([torch.from_numpy(y).cuda() for _, y in slots]).contiguous()
    refs = [ctrl.solve_batch(0.0, Qr[s], input_var=Yr[s]) for s in range(ring)]

    def run(nt):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            r = ctrl.resident_start(Qr, Yr, nt, timeout_s=3.0, ring_depth=ring,
                                    **(dict(integrate_dt=1e-3, max_speed=2.0) if integrate else {}))
            feeder = ctrl.resident_feed_stream()        # (one that makes progress beside the resident kernel)
            time.sleep(0.01)
            t0 = time.perf_counter()
            ctrl.resident_feed(r, nt, closed_loop=False, timeout_s=3.0, stream=feeder)
            r["stream"].synchronize()
            el = time.perf_counter() - t0
            feeder.synchronize()
            # (with the state kept by the kernel the rows move from tick to tick: that mode's parity is
            # tests/test_gpu_team.py::test_resident_ticks_integrate_the_state_and_take_streamed_targets)
            same = integrate or all(torch.equal(r["out"][s], refs[s][0]) and torch.equal(r["mode"][s], refs[s][2])
                                    for s in range(ring))
            if not (same and int(r["done"].min()) == nt):
                raise RuntimeError("resident ticks: an output slot differs from the launched tick on that slot's inputs")
            best = el if best is None else min(best, el)
        return best
    t_short, t_long = run(short), run(3 * short)
    per_tick = t_long / (3 * short)          # (the whole long run, its fixed ~2 ms included: the conservative figure)
    slope = (t_long - t_short) / (2 * short)
    return {
        "value": B / per_tick, "unit": "instance-steps/s", "ms_per_step": per_tick * 1e3,
        "config": {"workload": (("BASELINE config 2: %d x iiwa single pose task" if workload == "pose" else
                                 "BASELINE config 3: %d x iiwa priority stack") + " as RESIDENT ticks (one launch; device-side "
                                "tickets published ahead of the kernel; inputs and outputs in a ring of %d slots, a "
                                "different batch in each)%s") % (B, ring, "; the state kept by the kernel (q += clamp(dq, "
                               "+-2) * 1e-3 after every tick, clik_pinv_resident_run_state), only the targets read "
                               "from the ring" if integrate else ""), "batch_per_gpu": B, "ring_depth": ring,
                   "inputs": "%s seeds %s" % (dist_name, [seed + 17 * s for s in range(ring)]),
                   "kernel": ctrl.kernel_variant(B) + "/resident",
                   "timing": "one resident launch of %d ticks, host clock from the producer's launch to the kernel's exit "
                             "(best of 3), divided by the ticks; every output slot checked equal to a launched tick on "
                             "that slot's inputs; `slope_us_per_tick`: between that run and one of %d ticks (a run has a "
                             "fixed part of about 2 ms around its ticks)" % (3 * short, short),
                   "slope_us_per_tick": slope * 1e6,
                   "runs_us_per_tick": [t_short / short * 1e6, t_long / (3 * short) * 1e6]},
        "roofline": {"bound": "hbm", "achieved": 172.0 * B / per_tick / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": 172.0 * B / per_tick / 1e9 / 8000.0, "traffic": None, "tick_us": per_tick * 1e6},
        # (run() raises unless every output slot equals a launched tick on that slot's inputs, bit for bit)
        "check": {"ok": True, "what": "every ring slot bit-equal to a launched tick on its inputs (which tests hold to the oracle)"},
    }


def measure_resident_qp(fk, dist_name, seed, B=16384 - 64, short=10000, ring=4):
    """BASELINE config 4 through clik_qp_resident_run over a ring of `ring` slots, every ticket published ahead: after
    the first tick every tick is hot-started from the working set the kernel keeps (the steady state of a control
    loop).  16320 instances: the kernel takes a SIMD's whole register file and leaves one CU to the ticket feeder."""
    import torch
    import casclik_amd as cc
    from casclik_amd import skills
    ctrl = cc.ReactiveQPController(skill_spec=skills.qp_skill(fk))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    # (the same instances in every slot - a hot start means something only when successive ticks belong to the same
    # instances' loop - with the TARGET moved from slot to slot, a millimetre per tick along x, y, z in turn, as a
    # trajectory does: the hot starts then find working sets that are close but not all optimal)
    import numpy as np
    q_one, y_one = skills.synthetic_inputs(fk, B, seed=seed, distribution=dist_name)
    slots = []
    for s_ in range(ring):
        y_s = y_one.copy()
        y_s[:, s_ % 3] += 1e-3 * s_
        slots.append((q_one, y_s))
    Qr = torch.stack([torch.from_numpy(q).cuda() for q, _ in slots]).contiguous()
    Yr = torch.stack([torch.from_numpy(y).cuda() for _, y in slots]).contiguous()
    refs = [ctrl.solve_batch(0.0, Qr[s], input_var=Yr[s], use_hot=False) for s in range(ring)]

    def run(nt):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            r = ctrl.resident_start(Qr, Yr, nt, timeout_s=3.0, ring_depth=ring)
            feeder = ctrl.resident_feed_stream()        # (one that makes progress beside the resident kernel)
            time.sleep(0.01)
            t0 = time.perf_counter()
            ctrl.resident_feed(r, nt, closed_loop=False, timeout_s=3.0, stream=feeder)
            r["stream"].synchronize()
            el = time.perf_counter() - t0
            feeder.synchronize()
            same = all(torch.equal(r["status"][s], refs[s][3]) and
                       float((r["out"][s] - refs[s][0]).abs().nan_to_num(0.0).max()) < 1e-6 for s in range(ring))
            if not (same and int(r["done"].min()) == nt):
                raise RuntimeError("resident QP ticks: an output slot differs from the launched tick on that slot's inputs")
            best = el if best is None else min(best, el)
        return best
    t_short, t_long = run(short), run(3 * short)
    per_tick = t_long / (3 * short)
    return {
        "value": B / per_tick, "unit": "instance-steps/s", "ms_per_step": per_tick * 1e3,
        "config": {"workload": "BASELINE config 4: %d x iiwa ReactiveQPController as RESIDENT ticks (one launch; tickets published "
                               "ahead; ring of %d slots holding the same instances with the target moved 1 mm from slot to slot; the "
                               "working sets stay in the kernel: hot-started from tick 2 on)" % (B, ring),
                   "batch_per_gpu": B, "ring_depth": ring, "kernel": ctrl.kernel_variant(B, hot=True) + "/resident",
                   "slope_us_per_tick": (t_long - t_short) / (2 * short) * 1e6,
                   "runs_us_per_tick": [t_short / short * 1e6, t_long / (3 * short) * 1e6]},
        "roofline": {"bound": "hbm", "achieved": 220.0 * B / per_tick / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": 220.0 * B / per_tick / 1e9 / 8000.0, "traffic": None, "tick_us": per_tick * 1e6},
        "check": {"ok": True, "what": "every ring slot: statuses equal and minimisers within 1e-6 of a cold launched tick on its inputs"},
    }


def compact_roofline(r):
    if not r:
        return r
    f64 = r.get("fp64") or {}
    out = {"bound": r["bound"], "achieved": _r(r.get("achieved"), 1), "peak": r["peak"], "unit": r["unit"],
           "frac": _r(r.get("frac"), 4), "traffic": r.get("traffic"),
           "traffic_source": (r.get("traffic_source") or "").replace("profiles/", "").split("_k")[0] or None,
           "tick_us": _r(r.get("tick_us"), 3), "kernel_body_us": _r(r.get("kernel_body_us"), 3),
           "fp64_frac": _r(f64.get("frac"), 3), "valu_issue_frac": _r(f64.get("valu_issue_frac"), 3),
           "binds": r.get("binds"), "bytes_per_instance": r.get("algorithmic_bytes_per_instance"),
           "issue_probe_body_us": (r.get("issue_probe") or {}).get("probe_body_us"),
           "measured": ["achieved", "frac", "tick_us"],
           "recorded": [k for k, v in (("traffic", r.get("traffic")), ("fp64_frac (flops)", f64.get("frac")),
                                       ("kernel_body_us (boundary)", r.get("kernel_body_us")),
                                       ("issue_probe_body_us", (r.get("issue_probe") or {}).get("probe_body_us"))) if v is not None],
           "body_source": (r.get("kernel_body_source") or "").replace("profiles/", "") or None}
    return {k: v for k, v in out.items() if v is not None or k in ("traffic", "frac", "achieved")}


def _r(v, n):
    return None if v is None else round(float(v), n)


def compact_cpu(c):
    if not c or "error" in c:
        return c
    return {"value": _r(c["value"], 0), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
            "quota_cores": c.get("cgroup_cpu_quota_cores"), "single_core": _r(c["single_core"]["value"], 0),
            "sample": "%d ticks x %s instances, C restatement (oracle/clik_oracle_c.c), OpenMP, sustained median"
                      % (c["repeats"], c["sample"].split(" of a ")[1].split("-instance")[0])}


def compact_entry(e):
    """an `extras` entry in a few numbers: us per tick, G instance-steps/s, kernel, roofline fractions, CPU port rate"""
    if "error" in e:
        return {"error": e["error"][:120]}
    r = e.get("roofline") or {}
    f64 = r.get("fp64") or {}
    out = {"us": _r(e["ms_per_step"] * 1e3, 3), "G_per_s": _r(e["value"] / 1e9, 3),
           "kernel": str(e["config"].get("kernel")).replace("qp_static_kQpPoseIiwa", "kQp").replace("Iiwa", ""),
           "hbm_frac": _r(r.get("frac"), 4), "fp64_frac": _r(f64.get("frac"), 3), "traffic": r.get("traffic")}
    if r.get("traffic_source"):
        pass        # (the counter case is the entry's own name, workload_inputs_B<batch>, in profiles/r6_counters.json)
    else:
        # (no counter pass for this entry - the rollouts, the resident ticks: no traffic / fp64 figures)
        out = {k: v for k, v in out.items() if v is not None}
    if "check" in e:
        out["ok"] = bool(e["check"].get("ok"))
    c = e.get("cpu_baseline")
    if c and "value" in c:
        out["cpu_M_per_s"] = _r(c["value"] / 1e6, 2)
    if "allgather" in e:
        out["allgather"] = e["allgather"]
    return out


NOTES = ("us = wall per tick; roofline.achieved = ALGORITHMIC bytes (172 B/instance-step, QP 220) / tick vs 8 TB/s, not DRAM traffic "
         "(ticks rotate through --ring buffers; `traffic` = PMC bytes per launch, RECORDED in profiles/r6_counters.json); fp64_frac: executed "
         "flops (PMC) vs 78.6 TF; cpu_M_per_s: C port on cpu_baseline.cores threads (quota_cores: cgroup), a baseline not a "
         "speed-up; cost model: profiles/r5_tick_cost_model.md, r6_load_placement.md; --full 1: details")


def device_uuid(index):
    """the device's UUID (so that "N ranks on N distinct GPUs" can be checked from the line alone)"""
    import torch
    try:
        return str(torch.cuda.get_device_properties(index).uuid)
    except Exception:
        try:
            out = subprocess.run(["rocm-smi", "-d", str(index), "--showuniqueid"], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, timeout=20).stdout.decode()
            for line in out.splitlines():
                if "Unique ID" in line:
                    return line.split(":")[-1].strip()
        except Exception:
            pass
    return "unknown"


def init_ranks(world, rank, dev, shared_gpu):
    """process group for the barrier / max-over-ranks of the timing protocol: RCCL ("nccl" backend), and if that cannot
    be brought up, gloo on the host - the compute lines must come out either way; what failed is carried in the line"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    errors = {}
    # (the communication libraries print their greetings on the process's stdout - "[Gloo] Rank 0 is connected to ..." -;
    # stdout belongs to the ONE JSON line: send file descriptor 1 to stderr while the group comes up)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        return _init_ranks(dist, world, rank, dev, shared_gpu, errors)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)


def _init_ranks(dist, world, rank, dev, shared_gpu, errors):
    if not shared_gpu:
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            import torch
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                          # the first collective is where RCCL really starts
            torch.cuda.synchronize()
            return dist, "nccl", errors
        except Exception as exc:
            errors["nccl_init"] = repr(exc)[:300]
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return dist, "gloo", errors
    except Exception as exc:
        errors["gloo_init"] = repr(exc)[:300]
        return None, None, errors


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.lanes:
        os.environ["CLIK_LANES"] = str(args.lanes)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    shared_gpu = os.environ.get("CLIK_BENCH_SHARED_GPU", "0") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, backend, dist_errors = None, None, {}
    ranks_seen, devices = 1, ["cuda:%d %s uuid %s" % (local_rank, torch.cuda.get_device_name(local_rank), device_uuid(local_rank))]
    if world > 1:
        phase("rank %d: process group" % rank)
        dist, backend, dist_errors = init_ranks(world, rank, dev, shared_gpu)
        if dist is not None:
            ranks_seen = dist.get_world_size()
            gathered = [None] * world
            dist.all_gather_object(gathered, "rank %d pid %d %s" % (rank, os.getpid(), devices[0]))
            devices = gathered
    ctx = Ctx(rank, world, dev, dist, shared_gpu or backend == "gloo")
    ctx.gpu_collectives = backend == "nccl"

    from casclik_amd import skills
    fk = skills.iiwa()
    phase("headline: %s, %d instances per GPU, %d GPU(s)" % (args.workload, args.batch, world))
    head, (spec, opts, Q, Y) = measure(
        ctx, fk, args.workload, args.batch, args.dist, args.seed, args.steps, args.warmup,
        TPL=args.ticks_per_launch, qp_hot=args.qp_hot, use_graph=args.graph, ramp_ms=args.ramp_ms,
        min_timed_ms=args.min_timed_ms, replays=args.replays, allgather=args.allgather or (1 if world > 1 else 0),
        global_batch=args.global_batch, ring=args.ring)

    head_cpu = None
    if rank == 0 and args.cpu_baseline and world == 1:
        phase("cpu_baseline (C restatement on the host cores)")
        try:
            head_cpu = cpu_baseline(args.workload, spec, opts, Q, Y, args.cpu_seconds)
        except Exception as exc:            # the baseline must never sink the bench line
            head_cpu = {"error": repr(exc)}

    default_headline = (args.workload == "stack" and args.batch == 16384 and args.ticks_per_launch == 1
                        and not args.global_batch and not args.qp_hot and not args.lanes and args.graph == 1)
    want_extras = args.extras == 1 or (args.extras == -1 and default_headline and world == 1)
    extras = []
    if default_headline and world > 1 and args.extras != 0:
        # N > 1: next to the weak line (16384 instances per GPU: a tick is one wave per SIMD long whatever N is), the
        # STRONG line BASELINE config 5 names (ONE batch of 131072 cut over the ranks) and the regime in which a GPU is
        # worth a GPU (131072 instances PER GPU); the all-gather of dq is timed apart in each
        for name, kw in (("strong_B%d" % args.strong_batch, dict(global_batch=args.strong_batch) if args.strong_batch else None),
                         ("weak_B%d_per_gpu" % args.big_batch, dict(B=args.big_batch) if args.big_batch else None)):
            if kw is None:
                continue
            phase(name)
            try:
                ent, _ = measure(ctx, fk, "stack", kw.get("B", args.batch), args.dist, args.seed, args.steps, args.warmup,
                                 ramp_ms=100.0, min_timed_ms=max(args.extras_timed_ms, 800.0), allgather=1,
                                 global_batch=kw.get("global_batch", 0), ring=args.ring)
                extras.append(dict({"name": name, "n_gpus": world, "dtype": "f64",
                                    "scaling": "strong" if "global_batch" in kw else "weak"}, **ent))
            except Exception as exc:
                extras.append({"name": name, "error": repr(exc)})
    if want_extras:
        # the other BASELINE configurations under the same clock; short brackets
        # (config 4 twice: cold, and hot-started from the tick's own working set - the reference's qpOASES instance
        # hot-starts every solve after the first, reactive_qp.py:491-513: the steady state of a control loop)
        # ... and the on-device rollouts of both controllers (solve -> clamp -> integrate, K ticks per launch: the
        # notebooks' simulation loops; `roofline.frac` is null for them)
        # (config 4 also at 4096 instances: cold ticks of batches below one block per CU run four waves per 64 instances,
        # each with its own start of the active-set passes - clik_qp_static.hpp, FOLIO)
        for (wl, b, hot, tpl) in (("pose", 4096, 0, 1), ("qp", 16384, 0, 1), ("pose", 16384, 0, 1), ("qp", 16384, 1, 1),
                                  ("qp", 16384, 2, 1), ("qp", 4096, 0, 1),
                                  ("stack", 131072, 0, 1), ("qp", 131072, 0, 1), ("stack", 1048576, 0, 1),
                                  ("stack", 16384, 0, 256), ("qp", 16384, 0, 64)):
            up = lambda v: -(-max(v, tpl) // tpl) * tpl              # noqa: E731  (a whole number of launches)
            name = "%s_B%d%s%s" % (wl, b, ("_hot_settled" if hot == 2 else "_hot") if hot else "",
                                   "_rollout%d" % tpl if tpl > 1 else "")
            phase(name)
            try:
                ent, (sp, op, q_, y_) = measure(ctx, fk, wl, b, args.dist, args.seed,
                                                up(max(min(args.steps, 200) if b > 200000 else args.steps, 8 * tpl)),
                                                up(min(args.warmup, 20) if b > 200000 else args.warmup), TPL=tpl, qp_hot=hot,
                                                ramp_ms=100.0, min_timed_ms=args.extras_timed_ms,
                                                replays=8 if b > 200000 else 0, ring=args.ring)
            except Exception as exc:        # (an extra must never cost the headline its line)
                extras.append({"name": name, "error": repr(exc)})
                continue
            ent = dict({"name": name, "n_gpus": world, "dtype": "f64"}, **ent)
            if rank == 0 and args.cpu_baseline and world == 1 and not hot and tpl == 1 and b <= 131072:
                try:
                    ent["cpu_baseline"] = cpu_baseline(wl, sp, op, q_, y_, max(1.0, args.cpu_seconds / 3))
                except Exception as exc:
                    ent["cpu_baseline"] = {"error": repr(exc)}
            extras.append(ent)

        # config 3 as RESIDENT ticks (one launch, device-side tickets, all published ahead: include/clik.h)
        for name, integ, wl, b in (("stack_B16384_resident_fed_ahead", False, "stack", 16384),
                                   ("stack_B16384_resident_state_in_kernel", True, "stack", 16384),
                                   ("pose_B4096_resident_fed_ahead", False, "pose", 4096),
                                   ("pose_B16384_resident_fed_ahead", False, "pose", 16384)):
            phase(name)
            try:
                extras.append(dict({"name": name, "n_gpus": world, "dtype": "f64"},
                                   **measure_resident(fk, args.dist, args.seed, B=b, integrate=integ, workload=wl)))
            except Exception as exc:
                extras.append({"name": name, "error": repr(exc)})

    if want_extras:
        # (16320 instances: the resident QP kernel takes a SIMD's whole register file and leaves one CU to the feeder)
        phase("qp_B16320_resident_fed_ahead")
        try:
            extras.append(dict({"name": "qp_B16320_resident_fed_ahead", "n_gpus": world, "dtype": "f64"},
                               **measure_resident_qp(fk, args.dist, args.seed)))
        except Exception as exc:
            extras.append({"name": "qp_B16320_resident_fed_ahead", "error": repr(exc)})
    if rank == 0:
        phase("done")
        out = {
            "metric": METRIC,
            "value": head["value"], "unit": "instance-steps/s", "n_gpus": world, "steps": head["config"]["timed_steps"],
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": head["config"], "roofline": head["roofline"],
            "nranks_seen": ranks_seen, "devices": devices,
        }
        out["config"]["steps_per_graph_arg"] = args.steps
        if world > 1:
            out["process_group"] = backend
            if dist_errors:
                out["process_group_errors"] = dist_errors
        if "allgather" in head:
            out["allgather"] = head["allgather"]
        out["cpu_baseline"] = head_cpu
        if "check" in head:
            out["check"] = head["check"]
        if world > 1:
            # how to read the curve the driver builds from the per-N lines
            out["scaling_note"] = ("weak line: 16384 instances PER GPU - a tick is one wave per SIMD long whatever N is, so "
                                   "value scales ~N by construction (shards are independent, no data-path collective). "
                                   "The STRONG line of BASELINE config 5 is extras.strong_B131072: one GPU runs 131072 "
                                   "instances in ~9.6 us, eight GPUs at 16384 each take ~3.7 us - expect ~2.6x from 8 GPUs "
                                   "on that batch, not 8x; extras.weak_B131072_per_gpu is the regime where every GPU is full")
        if extras:
            out["extras"] = extras
        if not args.full:
            c = out["config"]
            out = dict({"notes": NOTES}, **out)
            out["config"] = {"workload": c["workload"], "batch_per_gpu": c["batch_per_gpu"], "inputs": c["inputs"],
                             "kernel": c["kernel"], "launch": c["launch"], "replays": c["replays"],
                             "steps_per_graph_arg": args.steps, "timed_ms": _r(c["timed_ms"], 1),
                             "parallelism": c["parallelism"]}
            if "global_batch" in c:
                out["config"]["global_batch"] = c["global_batch"]
            out["roofline"] = compact_roofline(out["roofline"])
            if "check" in out:
                out["check"] = {k: out["check"].get(k) for k in ("ok", "instances", "rule", "worst_err_over_tol", "error")
                                if k in out["check"]}
            out["cpu_baseline"] = compact_cpu(head_cpu)
            out["devices"] = [d.split(" pid ")[0] + " " + d.split(" ", 4)[-1] if " pid " in d else d for d in devices][:8]
            uu = [d.split(" uuid ")[-1] for d in devices if " uuid " in d]
            out["distinct_gpus"] = len(set(uu)) if uu and "unknown" not in uu else None
            if extras:
                out["extras"] = {e["name"]: compact_entry(e) for e in extras}
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
