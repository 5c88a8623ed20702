#!/usr/bin/env python3
"""How close the HIP kernels come to the STATED tolerance rule (tests/tolerances.py: err <= max(FLOOR, FACTOR u kappa)):
per case the worst err / (u kappa) over its instances - the constant the rule's FACTOR has to cover - for
  (a) every fixture recorded from the reference's own Python (tests/golden/ref_pins.npz), HIP against the REFERENCE RUN;
  (b) the BASELINE skills on mixed and near-singular inputs, every kernel family a batch size selects;
  (c) random members of the config-3 family at damping 1e-9 ... 1e-5 (tools/fuzz_team.py's generator);
  (d) the QP fixtures and BASELINE config 4 (kappa = clik_oracle.qp_condition).
    python tools/tolerance_sweep.py [n_random = 24] > profiles/r4_tolerance_sweep.txt      (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                   # noqa: E402

import casclik_amd as cc                             # noqa: E402
from casclik_amd import skills, sym as cs            # noqa: E402
from oracle import clik_oracle                       # noqa: E402
import refpins                                       # noqa: E402
from tolerances import U, FACTOR, FLOOR, ILL_POSED, rel_err, rtol_from_cond    # noqa: E402

n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 24
overall = {"pinv": 0.0, "qp": 0.0}


def line(kind, name, err, kappa, posed, extra=""):
    ratio = float((err[posed] / (U * kappa[posed])).max()) if posed.any() else 0.0
    over_floor = float((err[posed] / np.maximum(FLOOR, FACTOR * U * kappa[posed])).max()) if posed.any() else 0.0
    overall[kind] = max(overall[kind], over_floor)
    print("%-5s %-44s n %5d  kappa %.1e .. %.1e  err %.2e  err/(u kappa) %6.3f  err/tol %5.3f  ill-posed %d %s" % (
        kind, name, len(err), kappa.min(), kappa.max(), err[posed].max() if posed.any() else 0.0, ratio, over_floor,
        int((~posed).sum()), extra))


# ---- (a) reference-run fixtures ---------------------------------------------------------------------------------------
for name in refpins.PINV_NAMES:
    built = refpins.product_skill(name)
    Q, Y, X, times = refpins.arrays(name)
    ctrl = cc.PseudoInverseController(skill_spec=built["spec"], options=dict(built["options"]) or None)
    ctrl.setup_problem_functions()
    for ti, t in enumerate(times):
        kappa = np.zeros(len(Q))
        _, rmode = clik_oracle.pinv_solve_batch(built["spec"], built["options"] or None, float(t), Q, Y=Y, cond_out=kappa)
        dq, _, mode = ctrl.solve_batch(float(t), Q, input_var=Y)
        err = rel_err(dq, refpins.PINS[name + "_dq"][ti])
        line("pinv", "%s t=%.1f (%s)" % (name, t, ctrl.kernel_variant(len(Q))[-18:]), err, kappa,
             rtol_from_cond(kappa) < ILL_POSED, "modes equal %s" % bool(np.array_equal(mode, refpins.PINS[name + "_mode"][ti])))
for name in refpins.QP_NAMES:
    built = refpins.product_skill(name)
    Q, Y, X, times = refpins.arrays(name)
    ctrl = cc.ReactiveQPController(skill_spec=built["spec"], options=dict(built["options"]) or None)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    kappa = np.ones(len(Q))
    _, _, _, rst = clik_oracle.qp_solve_batch(built["spec"], float(times[0]), Q, X=X, Y=Y, cond_out=kappa)
    dq, dx, slack, status = ctrl.solve_batch(float(times[0]), Q, input_var=Y, virtual_var=X)
    ok = (refpins.ref_status(name) == 0) & (status == 0)
    got = np.hstack([a for a in (dq, dx, slack) if a is not None and np.ndim(a) == 2 and a.shape[1]])
    want = np.hstack([refpins.PINS[name + k] for k in ("_dq", "_dx", "_slack") if refpins.PINS[name + k].shape[1]])
    err = rel_err(got[ok], want[ok])
    line("qp", "%s (%s)" % (name, ctrl.kernel_name[:20]), err, kappa[ok], rtol_from_cond(kappa[ok]) < ILL_POSED,
         "status equal %s" % bool(np.array_equal(status, refpins.ref_status(name))))

# ---- (b) BASELINE skills -----------------------------------------------------------------------------------------------
fk = skills.iiwa()
rng = np.random.default_rng(0)
for label, spec, opts in (("config 3 stack", skills.stack_skill(fk), dict(skills.STACK_OPTIONS)),
                          ("config 2 pose", skills.pose_skill(fk), None), ("config 1 position", skills.position_skill(fk), None)):
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
    ctrl.setup_problem_functions()
    for dist, B in (("mixed", 1024), ("interior", 1024), ("near-singular", 512)):
        Q, Y = skills.synthetic_inputs(fk, B, seed=3, distribution="mixed" if dist == "near-singular" else dist)
        if dist == "near-singular":
            Q = rng.normal(0.0, 1e-3, size=Q.shape)                   # the stretched-out arm
        Yk = Y[:, :spec.n_input_var] if spec.n_input_var else None
        kappa = np.zeros(B)
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, 0.0, Q, Y=Yk, cond_out=kappa)
        for reps in (1, 20, 40):            # the same instances inside batches that select the other kernel families
            Qb, Yb = np.tile(Q, (reps, 1)), (None if Yk is None else np.tile(Yk, (reps, 1)))
            dq, _, mode = ctrl.solve_batch(0.0, Qb, input_var=Yb)
            err = rel_err(dq[:B], ref)
            same = mode[:B] == rmode
            line("pinv", "%s %s B=%d (%s)" % (label, dist, len(Qb), ctrl.kernel_variant(len(Qb))[-16:]), err[same], kappa[same],
                 rtol_from_cond(kappa[same]) < ILL_POSED, "modes differ %d" % int((~same).sum()))
qspec = skills.qp_skill(fk)
qctrl = cc.ReactiveQPController(skill_spec=qspec)
qctrl.setup_problem_functions()
qctrl.setup_solver()
for dist in ("mixed", "interior"):
    Q, Y = skills.synthetic_inputs(fk, 768, seed=4, distribution=dist)
    kappa = np.ones(len(Q))
    rdq, _, rsl, rst = clik_oracle.qp_solve_batch(qspec, 0.0, Q, Y=Y, cond_out=kappa)
    dq, _, sl, st = qctrl.solve_batch(0.0, Q, input_var=Y)
    ok = (rst == 0) & (st == 0)
    err = rel_err(np.hstack([dq, sl])[ok], np.hstack([rdq, rsl])[ok])
    line("qp", "config 4 %s (%s)" % (dist, qctrl.kernel_name[:20]), err, kappa[ok], rtol_from_cond(kappa[ok]) < ILL_POSED,
         "status differ %d" % int((rst != st).sum()))

# ---- (c) random members of the config-3 family, damping 1e-9 ... 1e-5 ---------------------------------------------------
FK = {"iiwa": fk, "ur5": skills.ur5()}
rng = np.random.default_rng(11)
for s in range(n_random):
    robot = "ur5" if rng.random() < 0.4 else "iiwa"
    f = FK[robot]
    n = len(f["joint_names"])
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7 + n)
    T = f["T_fk"](q)
    lo, hi = np.array(f["lower"]), np.array(f["upper"])
    scale = rng.uniform(0.5, 1.0)
    limits = cc.SetConstraint(label="limits", expression=q, priority=0, set_max=scale * hi, set_min=scale * lo)
    m = 6 if rng.random() < 0.7 else 3
    expr = skills._pose_expression(T, y) if m == 6 else T[:3, 3] - y[:3]
    K = float(rng.uniform(1.0, 12.0)) if rng.random() < 0.6 else np.diag(rng.uniform(1.0, 10.0, size=m)) + 0.3 * rng.normal(size=(m, m))
    pose = cc.EqualityConstraint("task", expr, gain=K, constraint_type="soft", priority=1)
    js = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
    third = cc.EqualityConstraint("joints", cs.vertcat(*[q[j] - y[7 + j] for j in js]), gain=float(rng.uniform(0.2, 3.0)),
                                  constraint_type="soft", priority=2)
    spec = cc.SkillSpecification("fuzz_team", t, q, input_var=y, constraints=[third, limits, pose])
    lam = float(10 ** rng.uniform(-9, -5))
    opts = {"multidim_sets": True, "feedforward": True, "damping_factor": lam}
    B = 256
    Q, Y7 = skills.synthetic_inputs(f, B, seed=int(rng.integers(1 << 30)), distribution="mixed")
    if rng.random() < 0.3:
        Q[: B // 4] = rng.normal(0.0, 1e-4, size=(B // 4, n))
    Y = np.hstack([Y7, rng.uniform(0.3 * lo, 0.3 * hi, size=(B, n))])
    kappa = np.zeros(B)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, 0.3, Q, Y=Y, cond_out=kappa)
    for values in ("1", "0"):
        os.environ["CLIK_JIT_VALUES"] = values
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
        ctrl.setup_problem_functions()
        dq, _, mode = ctrl.solve_batch(0.3, Q, input_var=Y)
        same = mode == rmode
        line("pinv", "random %2d %-4s m=%d lam=%.0e (%s)" % (s, robot, m, lam, ctrl.kernel_variant(B)[-8:]), rel_err(dq, ref)[same],
             kappa[same], rtol_from_cond(kappa[same]) < ILL_POSED, "modes differ %d" % int((~same).sum()))
    os.environ.pop("CLIK_JIT_VALUES", None)
print("worst err / tol of the rule (FACTOR %g, FLOOR %g): pinv %.3f, qp %.3f" % (FACTOR, FLOOR, overall["pinv"], overall["qp"]))
