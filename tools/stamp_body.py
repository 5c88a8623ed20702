#!/usr/bin/env python3
"""Kernel BODY time of the headline kernels: first wave start -> last wave end of one launch, from s_memrealtime
stamps (the 100 MHz constant clock shared by all XCDs; 10 ns resolution) every wave takes when it starts and after
its last store has landed (-DCLIK_BODY_STAMPS instantiation, never the shipped kernels).  For every sample a graph of
G back-to-back ticks is replayed and the stamps of its LAST launch are read back; S samples give median / p10 / p90.
Next to it: the per-tick wall time of the same graph (HIP events), i.e. body + launch boundary, and the un-stamped
kernel's tick time for the stamps' own overhead.

    python tools/stamp_body.py [samples=1000] > gpurun_out/.../body.log     (writes gpurun_out/r6body/r6_body_time.json/.csv; copied to profiles/)
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402

LIGHT = "--light" in sys.argv       # -DCLIK_BODY_STAMPS=2: block 0 stamps the start, every eighth block the end (see the macro)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
S = int(args[0]) if args else 1000
G = 64
OUT = {}


def tick_time_us(tick, n=1024, reps=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tick()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            tick()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def measure(name, workload, B, lanes, key_kernel, hot=False):
    import importlib
    os.environ["CLIK_JIT_DEFINES"] = ""
    os.environ.pop("CLIK_LANES", None)
    if lanes:
        os.environ["CLIK_LANES"] = str(lanes)
    import casclik_amd as cc
    from casclik_amd import skills, jit
    fk = skills.iiwa()

    def make():
        if workload == "qp":
            c = cc.ReactiveQPController(skill_spec=skills.qp_skill(fk))
        elif workload == "pose":
            c = cc.PseudoInverseController(skill_spec=skills.pose_skill(fk))
        else:
            c = cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS))
        c.setup_problem_functions()
        c.setup_solver()
        return c

    Q, Y = skills.synthetic_inputs(fk, B, seed=0, distribution="mixed")
    Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
    plain = make()
    kernel = plain.kernel_variant(B, hot=True) if hot else plain.kernel_variant(B)
    if kernel != key_kernel:
        print("(%s: the library serves this configuration with %s, not %s)" % (name, kernel, key_kernel))
    bind_kw = dict(hot_start=True) if hot else {}
    t_plain = tick_time_us(plain.bind_batch(Qd, input_var=Yd, **bind_kw))
    os.environ["CLIK_JIT_DEFINES"] = "-DCLIK_BODY_STAMPS=2" if LIGHT else "-DCLIK_BODY_STAMPS"
    ctrl = make()
    lib = (jit.attach_qp_values if workload == "qp" else jit.attach_values).last_library
    tick = ctrl.bind_batch(Qd, input_var=Yd, **bind_kw)
    t_stamped = tick_time_us(tick)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tick()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(G):
            tick()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    waves = min(32768, (B * (4 if ("team4" in kernel or "quadv" in kernel or "folio4" in kernel) else 1) + 63) // 64)
    buf = (C.c_ulonglong * (2 * waves))()
    lib.clik_jit_read_body.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    body, spread_start, per_wave = [], [], []
    for _ in range(S):
        g.replay()
        torch.cuda.synchronize()
        assert lib.clik_jit_read_body(buf, 2 * waves) == 0
        st = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(-1, 2)
        if LIGHT:
            starts, ends = st[:, 0][st[:, 0] > 0], st[:, 1][st[:, 1] > 0]
            # (the kernels of one shared object share the stamp array: slots another kernel wrote seconds ago are stale)
            starts, ends = starts[starts > ends.max() - 100000], ends[ends > ends.max() - 100000]
            body.append((ends.max() - starts.min()) * 0.01)
            spread_start.append((starts.max() - starts.min()) * 0.01)
            both = (st[:, 0] > 0) & (st[:, 1] > 0)
            per_wave.append(np.median((st[:, 1] - st[:, 0])[both]) * 0.01 if both.any() else float("nan"))
            continue
        body.append((st[:, 1].max() - st[:, 0].min()) * 0.01)             # 100 MHz ticks -> us
        spread_start.append((st[:, 0].max() - st[:, 0].min()) * 0.01)
        per_wave.append(np.median(st[:, 1] - st[:, 0]) * 0.01)
    body, spread_start, per_wave = np.array(body), np.array(spread_start), np.array(per_wave)
    ent = {"kernel": kernel, "batch": B, "waves": int(waves), "samples": S, "graph_ticks": G,
           "body_us_median": float(np.median(body)), "body_us_p10": float(np.percentile(body, 10)),
           "body_us_p90": float(np.percentile(body, 90)),
           "wave_lifetime_us_median": float(np.median(per_wave)),
           "first_to_last_wave_start_us_median": float(np.median(spread_start)),
           "tick_us_stamped_build": t_stamped, "tick_us_shipped_build": t_plain,
           "boundary_us": t_stamped - float(np.median(body)),
           "method": "s_memrealtime (100 MHz, 10 ns) per wave at entry and after s_waitcnt vmcnt(0) at exit; body = "
                     "max(end) - min(start) over the waves of the last launch of a %d-tick graph; tick_us = HIP events "
                     "around graph replays / ticks; boundary = tick (stamped build) - body" % G
                     + ("; LIGHT stamps: block 0 stamps the start, every eighth block and the last the end" if LIGHT else "")}
    OUT["%s_mixed_B%d_%s%s" % (workload, B, kernel, "_hot" if hot else "")] = ent
    print("%-34s body %.2f us (p10 %.2f, p90 %.2f)  wave lifetime %.2f  start spread %.2f   tick stamped %.3f / shipped "
          "%.3f us -> boundary %.2f us" % (name, ent["body_us_median"], ent["body_us_p10"], ent["body_us_p90"],
                                           ent["wave_lifetime_us_median"], ent["first_to_last_wave_start_us_median"],
                                           t_stamped, t_plain, ent["boundary_us"]))


# round 6: every kernel the bench line reports
measure("config 3, 16384, team4v", "stack", 16384, 0, "kStackIiwa/team4v")
measure("config 2, 4096, quadv", "pose", 4096, 0, "kPose6Iiwa/quadv")
measure("config 2, 16384, quadv", "pose", 16384, 0, "kPose6Iiwa/quadv")
measure("config 4, 16384, cold (folio4)", "qp", 16384, 0, "qp_static_kQpPoseIiwa/v/folio4")
measure("config 4, 16384, hot-started", "qp", 16384, 0, "qp_static_kQpPoseIiwa/v", hot=True)
measure("config 3, 131072, lanev", "stack", 131072, 0, "kStackIiwa/lanev")
measure("config 4, 131072", "qp", 131072, 0, "qp_static_kQpPoseIiwa/v")
TAG = "r6_body_time"
os.makedirs(os.path.join(ROOT, "gpurun_out", "r6body"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r6body", TAG + ".json"), "w") as f:
    json.dump(OUT, f, indent=1)
with open(os.path.join(ROOT, "gpurun_out", "r6body", TAG + ".csv"), "w") as f:
    f.write("key,kernel,batch,waves,samples,body_us_median,body_us_p10,body_us_p90,wave_lifetime_us_median,"
            "tick_us_stamped_build,tick_us_shipped_build,boundary_us\n")
    for k, e in OUT.items():
        f.write("%s,%s,%d,%d,%d,%.3f,%.3f,%.3f,%.3f,%.4f,%.4f,%.3f\n" % (
            k, e["kernel"], e["batch"], e["waves"], e["samples"], e["body_us_median"], e["body_us_p10"],
            e["body_us_p90"], e["wave_lifetime_us_median"], e["tick_us_stamped_build"], e["tick_us_shipped_build"],
            e["boundary_us"]))
