#!/usr/bin/env python3
"""Headline benchmark: CLIK steps/sec on the 7-DoF 3-task priority stack.

    python bench.py --gpus N --steps K --warmup W

A *step* is one controller tick over the whole batch resident in HBM (one
kernel launch per tick per GPU).  `value` = instance-steps per second =
(N * batch * steps timed) / time, the BASELINE.json metric "CLIK steps/sec (whole
node), 7-DoF 3-task priority stack, batch 16384" (batch is per GPU: weak
scaling, as in BASELINE config 5 = 131072 instances over 8 GPUs).

The JSON line also carries
  roofline      algorithmic HBM bytes (q 56 B + target 56 B +
                dq 56 B + mode 4 B per instance-step) over the kernel
                duration from HIP events on the launch stream, vs 8 TB/s
  cpu_baseline  the C restatement of the reference algorithm (oracle/, kind
                "port") timed on this host's cores on a bounded sample

Timing protocol: W untimed warm-up steps, then an untimed, time-based clock ramp
(--ramp-ms of replays: a fresh GPU needs ~100 ms of work before its clocks
settle, which a 20-step run never reaches), then R back-to-back replays of a
hipGraph inside ONE barrier + synchronize bracket.  The graph holds the K steps
repeated to >= 1024 ticks (a graph launch opens with a ~10 us device bubble that
a 20-tick graph would pay every 20 ticks).  `ms_per_step` = bracket wall time /
timed steps (max over ranks), `value` = instances * timed steps / wall;
`roofline.kernel_us` = median over the R replays of (HIP-event time of one
replay / its ticks).  R is chosen so that the bracket holds >= --min-timed-ms of
work and >= 50 replays; R and the timed steps are stated in `config`.

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
PROFILE_FILES = ("r2_counters.json", "r1_traffic.json")     # newest first


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=16384, help="instances per GPU")
    ap.add_argument("--workload", default="stack", choices=["stack", "pose", "qp"])
    ap.add_argument("--dist", default="mixed", choices=["interior", "mixed"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", type=int, default=1,
                    help="replay the K ticks from one hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--allgather", type=int, default=0,
                    help="also all-gather dq over RCCL every tick (reported separately)")
    ap.add_argument("--ticks-per-launch", type=int, default=1,
                    help="K > 1: the on-device rollout (K ticks of solve -> clamp -> Euler per launch, "
                         "SURVEY.md 8(d) 'launch-amortised'); steps must be a multiple of K")
    ap.add_argument("--qp-hot", type=int, default=0,
                    help="qp workload: carry each instance's working set from tick to tick (hot start)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="lanes per robot instance of the pinv kernel: 0 = the library's choice, 1 = "
                         "lane-per-instance kernels only, 4 / 8 / 16 = the multi-lane kernel (CLIK_LANES)")
    ap.add_argument("--ramp-ms", type=float, default=250.0, help="untimed clock ramp before the timed region")
    ap.add_argument("--min-timed-ms", type=float, default=60.0, help="least work inside the timed bracket")
    ap.add_argument("--replays", type=int, default=0, help="R (0: from --min-timed-ms, at least 50)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def self_launch(args):
    """--gpus N without a launcher: become the launcher.  Nothing here initialises a GPU
    (torch.cuda.device_count() only counts), and the ranks are fresh child processes."""
    import torch
    n = torch.cuda.device_count()
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


def cpu_baseline(workload, spec, opts, Q, Y, seconds):
    """Time the CPU restatement on the host cores this process may use.  Pseudo-inverse
    workloads: the C restatement (oracle/clik_oracle_c.c, OpenMP over instances; the thread
    count with the best throughput is reported).  QP: the numpy restatement (one core)."""
    import numpy as np
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    if workload == "qp":
        from oracle import clik_oracle
        n = min(len(Q), 256)
        t0 = time.perf_counter()
        clik_oracle.qp_solve_batch(spec, 0.0, Q[:n], Y=Y[:n])
        one = time.perf_counter() - t0
        reps = max(1, min(50, int(seconds / max(one, 1e-6)) - 1))
        t0 = time.perf_counter()
        for k in range(reps):
            lo = (k * n) % max(1, len(Q) - n + 1)
            clik_oracle.qp_solve_batch(spec, 0.0, Q[lo:lo + n], Y=Y[lo:lo + n])
        el = time.perf_counter() - t0
        return {"value": n * reps / el, "unit": "instance-steps/s", "cores": 1, "kind": "port",
                "host_cores_available": avail,
                "sample": "%d x %d instances of the bench batch, numpy restatement of reactive_qp.py:175-246 + dense "
                          "Goldfarb-Idnani (oracle/clik_oracle.py), one core" % (reps, n)}
    from oracle import c_oracle
    co = c_oracle.CPinvOracle(spec, opts)
    # enough rows per thread to amortise the OpenMP fork/join
    reps_rows = max(1, (8 * 16384) // len(Q))
    Qs, Ys = np.tile(Q, (reps_rows, 1)), np.tile(Y, (reps_rows, 1))
    sample = len(Qs)
    best_rate, best_threads = 0.0, 1
    for threads in sorted({1, min(8, avail), min(32, avail), min(64, avail), avail}):
        co.solve_batch(0.0, Qs, Y=Ys, nthreads=threads)
        t0 = time.perf_counter()
        co.solve_batch(0.0, Qs, Y=Ys, nthreads=threads)
        rate = sample / (time.perf_counter() - t0)
        if rate > best_rate:
            best_rate, best_threads = rate, threads
    one = sample / best_rate
    reps = max(1, min(5000, int(seconds / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        co.solve_batch(0.0, Qs, Y=Ys, nthreads=best_threads)
    el = time.perf_counter() - t0
    return {"value": sample * reps / el, "unit": "instance-steps/s", "cores": best_threads,
            "kind": "port", "host_cores_available": avail,
            "sample": "%d ticks of a %d-instance batch (the bench batch tiled %dx), C restatement of the "
                      "reference algorithm (oracle/clik_oracle_c.c), OpenMP over instances, best of "
                      "{1,8,32,64,all} threads" % (reps, sample, reps_rows)}


def profiled(workload, dist_name, batch, kernel):
    """Per-launch PMC figures of this configuration recorded under profiles/ (they cannot be collected
    from inside this process: separate rocprofv3 --pmc passes, tools/rocprof_counters.py).  Returns
    (entry, source) or (None, None): a configuration that was not profiled gets no traffic figure."""
    keys = ["%s_%s_B%d_%s" % (workload, dist_name, batch, kernel)]
    if str(kernel).endswith(("/lane", "/mp2")) or "/" not in str(kernel):
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

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from casclik_amd import skills
    fk = skills.iiwa()
    spec, opts, ctrl = make_workload(args.workload, fk)
    B = args.batch
    # every rank owns its own contiguous shard of the global batch (weak scaling)
    Q, Y = skills.synthetic_inputs(fk, B, seed=args.seed + 1000 * rank, distribution=args.dist)
    Qd = torch.from_numpy(Q).to(dev)
    Yd = torch.from_numpy(Y).to(dev)
    dQ = torch.empty((B, Q.shape[1]), dtype=torch.float64, device=dev)
    gathered = None
    if args.allgather and world > 1:
        gathered = torch.empty((world * B, Q.shape[1]), dtype=torch.float64, device=dev)

    TPL = args.ticks_per_launch
    if TPL > 1:
        if args.steps % TPL or args.warmup % TPL:
            raise SystemExit("--ticks-per-launch needs steps and warmup divisible by it")
        times = [0.0] * TPL
        state = {"q": Qd}

        def tick():         # one launch = TPL ticks; the state is carried from launch to launch
            state["q"] = ctrl.rollout_batch(times, state["q"], input_var=Yd, dt=1e-3, max_speed=2.0)[0]
        args.graph = 0
        args.steps //= TPL
        args.warmup //= TPL
    elif args.workload == "qp":
        tick = ctrl.bind_batch(Qd, input_var=Yd, out=dQ, hot_start=bool(args.qp_hot))
    else:
        tick = ctrl.bind_batch(Qd, input_var=Yd, out=dQ)

    def step():
        tick()
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, dQ)

    stream = torch.cuda.Stream(device=dev)
    K, W = args.steps, args.warmup
    with torch.cuda.stream(stream):
        for _ in range(W):
            step()
        stream.synchronize()
        graph = None
        # a replay = GK steps = the K steps repeated until the graph holds >= 1024 ticks: every graph launch
        # starts with a ~10 us bubble on the device (measured: 6.26 us per step from a 20-tick graph against
        # 5.74 us from a 2000-tick graph of the same kernel), which a control loop that enqueues ticks
        # continuously never sees
        GK = K * max(1, -(-1024 // K)) if (args.graph and gathered is None and TPL == 1) else K
        if args.graph and gathered is None:
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
                    step()

        # untimed clock ramp; its rate also sizes R
        t_r = time.perf_counter()
        n_ramp = 0
        while True:
            replay()
            n_ramp += 1
            if n_ramp % 4 == 0 or GK >= 500:
                stream.synchronize()
                if (time.perf_counter() - t_r) * 1e3 >= args.ramp_ms:
                    break
        stream.synchronize()
        est_ms = (time.perf_counter() - t_r) * 1e3 / n_ramp          # one replay, host-inclusive (upper bound)
        R = args.replays if args.replays > 0 else max(50, int(args.min_timed_ms / max(est_ms, 1e-3)) + 1)
        R = min(R, 20000)
        if dist is not None:
            rr = torch.tensor([R], dtype=torch.int64, device=dev)
            dist.all_reduce(rr, op=dist.ReduceOp.MAX)
            R = int(rr.item())
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record(stream)
        for r in range(R):
            replay()
            evs[r + 1].record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
    per_replay_ms = sorted(evs[r].elapsed_time(evs[r + 1]) for r in range(R))
    med_ms = per_replay_ms[R // 2]
    dev_ms = evs[0].elapsed_time(evs[R])
    if dist is not None:
        tt = torch.tensor([wall, dev_ms, med_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms, med_ms = float(tt[0]), float(tt[1]), float(tt[2])

    if rank == 0:
        K, W, GK = K * TPL, W * TPL, GK * TPL          # report in ticks
        timed_steps = R * GK
        total_steps = world * B * timed_steps
        value = total_steps / wall
        kern_us = med_ms * 1e3 / GK
        bytes_per_inst = 8 * (Q.shape[1] + Y.shape[1] + Q.shape[1]) + (0 if args.workload == "qp" else 4)
        alg_bytes = bytes_per_inst * B
        achieved = alg_bytes / (kern_us * 1e-6) / 1e9
        kernel = getattr(ctrl, "kernel_name", None)
        variant = getattr(ctrl, "kernel_variant", None)
        if callable(variant):
            kernel = variant(B)
        # (an on-device rollout moves its bytes once per launch of TPL ticks: no per-tick traffic figure was profiled)
        prof, prof_src = profiled(args.workload, args.dist, B, kernel) if TPL == 1 else (None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": prof.get("traffic_bytes") if prof else None,
                "traffic_source": prof_src if (prof and "traffic_bytes" in prof) else None,
                "kernel_us": kern_us, "kernel_us_mean": dev_ms * 1e3 / timed_steps,
                "algorithmic_bytes_per_launch": alg_bytes}
        if prof and "fp64_flops_per_launch" in prof:
            # executed fp64 flops (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 x 64 lanes, FMA = 2) over the measured
            # kernel time: what the VALUs did, not what the literal algorithm would need
            roof["fp64_valu_frac_executed"] = (prof["fp64_flops_per_launch"] / (kern_us * 1e-6)
                                               / (FP64_VALU_PEAK_TF * 1e12))
            roof["fp64_source"] = prof_src
        out = {
            "metric": "CLIK steps/sec (whole node), 7-DoF 3-task priority stack, batch 16384",
            "value": value, "unit": "instance-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / timed_steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": {"stack": "BASELINE config 3: %d x KUKA iiwa 7-DoF per GPU, priority stack "
                                      "[multidim joint-limit set; 6-D pose; joint centering], "
                                      "PseudoInverseController" % B,
                             "pose": "BASELINE config 2: %d x iiwa, single 6-D pose task" % B,
                             "qp": "BASELINE config 4: %d x iiwa ReactiveQPController%s" % (
                                 B, " (hot-started from the previous tick's working set)" if args.qp_hot else "")}[args.workload],
                "batch_per_gpu": B, "inputs": "%s seed %d" % (args.dist, args.seed),
                "kernel": kernel,
                "launch": ("hipGraph of %d ticks (the K ticks x %d)" % (GK, GK // K) if graph is not None
                           else "eager, one launch per tick") if TPL == 1
                          else "on-device rollout, %d ticks per launch (solve -> clamp -> Euler)" % TPL,
                "replays": R, "timed_steps": timed_steps, "clock_ramp_ms": args.ramp_ms,
                "timing": "R back-to-back replays in one barrier+synchronize bracket; "
                          "ms_per_step = wall / timed_steps, max over ranks",
                "ticks_per_s": timed_steps / wall,
                "parallelism": "dp%d (independent shards, no data-path collective)" % world,
            },
            "roofline": roof,
        }
        if args.cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, spec, opts, Q, Y, args.cpu_seconds)
            except Exception as exc:            # the baseline must never sink the bench line
                out["cpu_baseline"] = {"error": repr(exc)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
