#!/usr/bin/env python3
"""Random skills through the REFERENCE package and through the oracle (build container only: needs /root/reference).

tests/golden/ref_pins.npz holds 23 hand-written skills; this draws random ones - task kinds, priorities, gains (scalar,
vector-free matrix), 1-D and multidimensional sets on joints and on task-space coordinates, velocity constraints, a
virtual variable, time-dependent targets, every controller option - builds each TWICE from one recipe (the reference's
constraint classes and controllers over the stand-in casadi with kinematics multiplied out from the reference's URDF;
the product's front-end with its URDF converter), runs the reference's own `solve()` per instance and the numpy oracle
on the same inputs, and compares `print_constraints()` texts, modes, velocities, slack, QP data and statuses.

    python tools/fuzz_ref_vs_oracle.py [skills=40] [seed=0]          (summary: profiles/r3_fuzz_ref_vs_oracle.txt)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, GOLD)


def recipe(rng, n):
    """a skill as plain data (so that both sides build the same one)"""
    tasks = []
    kinds = ["position", "pose", "joints", "axis", "veleq", "set1d_joint", "set1d_task", "setmd_joint", "velset"]
    n_tasks = int(rng.integers(1, 5))
    used_sets = 0
    for k in range(n_tasks):
        kind = str(rng.choice(kinds))
        if kind in ("set1d_joint", "set1d_task", "setmd_joint") and used_sets >= 3:
            kind = "joints"
        task = {"kind": kind, "priority": int(rng.integers(0, 6)), "gain": float(rng.choice([0.5, 1.0, 3.0, 10.0])),
                "soft": bool(rng.random() < 0.6), "slack_weight": float(rng.choice([1.0, 5.0]))}
        if kind == "joints":
            m = int(rng.integers(1, n + 1))
            task["idx"] = sorted(rng.choice(n, size=m, replace=False).tolist())
            task["target"] = rng.uniform(-0.5, 0.5, m).tolist()
            if m > 1 and rng.random() < 0.3:
                A = rng.normal(size=(m, m))
                task["gain_matrix"] = (A.dot(A.T) / m + np.eye(m)).tolist()
        elif kind == "axis":
            task["row"], task["col"] = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            task["target"] = float(rng.uniform(-0.5, 0.5))
        elif kind == "veleq":
            task["idx"] = int(rng.integers(0, n))
            task["target"] = float(rng.uniform(-0.3, 0.3))
        elif kind == "set1d_joint":
            task["idx"] = int(rng.integers(0, n))
            task["scale"] = float(rng.choice([0.3, 0.6]))
            used_sets += 1
        elif kind == "set1d_task":
            task["axis"] = int(rng.integers(0, 3))
            task["lo"], task["hi"] = (-0.2, 0.5) if rng.random() < 0.5 else (0.1, 0.9)
            used_sets += 1
        elif kind == "setmd_joint":
            m = int(rng.integers(2, n + 1))
            task["idx"] = sorted(rng.choice(n, size=m, replace=False).tolist())
            task["scale"] = float(rng.choice([0.3, 0.6]))
            used_sets += 1
        elif kind == "velset":
            task["scale"] = float(rng.choice([0.2, 1.0]))
        task["time"] = bool(kind in ("position", "joints") and rng.random() < 0.3)
        tasks.append(task)
    if not any(t["kind"] in ("position", "pose", "joints", "axis", "veleq") for t in tasks):
        tasks.append({"kind": "position", "priority": 9, "gain": 2.0, "soft": True, "slack_weight": 1.0, "time": False})
    virtual = bool(rng.random() < 0.25)
    options = {"feedforward": bool(rng.random() < 0.8), "damping_factor": float(rng.choice([1e-7, 1e-4, 1e-2])),
               "multidim_sets": True, "converge_final_set_to_max": bool(rng.random() < 0.3)}
    return {"tasks": tasks, "virtual": virtual, "options": options, "p_des": rng.uniform(0.2, 0.6, 3).tolist(),
            "quat": (lambda v: (v / np.linalg.norm(v)).tolist())(rng.normal(size=4))}


def build(rec, cs, cc, T_fk, ori_err, lower, upper, vmax):
    n = len(lower)
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("dq", n)
    x, dx = (cs.MX.sym("x"), cs.MX.sym("dx")) if rec["virtual"] else (None, None)
    T = T_fk(q)
    cons = []
    for k, tk in enumerate(rec["tasks"]):
        kind = tk["kind"]
        ctype = "soft" if tk["soft"] else "hard"
        common = dict(label="%s_%d" % (kind, k), priority=tk["priority"])
        shift = (0.05 * cs.sin(0.7 * t)) if tk.get("time") else 0.0
        if kind == "position":
            p_des = np.asarray(rec["p_des"])
            expr = T[:3, 3] - p_des + (cs.vertcat(shift, 0.0, 0.02 * t) if tk.get("time") else 0.0)
            if rec["virtual"]:
                expr = expr + cs.vertcat(0.1 * cs.sin(0.5 * x), 0.0, 0.0)
            cons.append(cc.EqualityConstraint(expression=expr, gain=tk["gain"], constraint_type="soft",
                                              slack_weight=tk["slack_weight"], **common))
        elif kind == "pose":
            expr = cs.vertcat(T[:3, 3] - np.asarray(rec["p_des"]), ori_err(T[:3, :3], np.asarray(rec["quat"])))
            cons.append(cc.EqualityConstraint(expression=expr, gain=tk["gain"], constraint_type="soft",
                                              slack_weight=tk["slack_weight"], **common))
        elif kind == "joints":
            idx = tk["idx"]
            expr = cs.vertcat(*[q[j] for j in idx]) - np.asarray(tk["target"]) + shift
            gain = np.asarray(tk["gain_matrix"]) if "gain_matrix" in tk else tk["gain"]
            cons.append(cc.EqualityConstraint(expression=expr, gain=gain, constraint_type=ctype,
                                              slack_weight=tk["slack_weight"], **common))
        elif kind == "axis":
            cons.append(cc.EqualityConstraint(expression=T[tk["row"], tk["col"]] - tk["target"], gain=tk["gain"],
                                              constraint_type="soft", slack_weight=tk["slack_weight"], **common))
        elif kind == "veleq":
            cons.append(cc.VelocityEqualityConstraint(expression=q[tk["idx"]], target=tk["target"],
                                                      constraint_type="soft", **common))
        elif kind == "set1d_joint":
            j = tk["idx"]
            cons.append(cc.SetConstraint(expression=q[j], set_min=float(tk["scale"] * lower[j]),
                                         set_max=float(tk["scale"] * upper[j]), gain=tk["gain"], **common))
        elif kind == "set1d_task":
            cons.append(cc.SetConstraint(expression=T[tk["axis"], 3], set_min=tk["lo"], set_max=tk["hi"], gain=tk["gain"],
                                         constraint_type=ctype, **common))
        elif kind == "setmd_joint":
            idx = tk["idx"]
            cons.append(cc.SetConstraint(expression=cs.vertcat(*[q[j] for j in idx]), set_min=tk["scale"] * lower[idx],
                                         set_max=tk["scale"] * upper[idx], gain=tk["gain"], **common))
        elif kind == "velset":
            cons.append(cc.VelocitySetConstraint(expression=q, set_min=-tk["scale"] * vmax, set_max=tk["scale"] * vmax,
                                                 **common))
    if rec["virtual"]:
        cons.append(cc.EqualityConstraint(label="along", expression=3.0 - x, gain=0.5, constraint_type="soft", priority=7))
    kw = dict(virtual_var=x, virtual_vel_var=dx) if rec["virtual"] else {}
    return cc.SkillSpecification(label="fuzz", time_var=t, robot_var=q, robot_vel_var=dq, constraints=cons, **kw)


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tolerances import rtol_from_cond, ILL_POSED      # noqa: E402


def main():
    n_skills = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import importlib.util
    # the reference side: its package over the stand-in casadi, kinematics from its URDFs (make_ref_golden's helpers)
    spec_ = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(GOLD, "make_ref_golden.py"))
    gen = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(gen)
    rcs, rcc = gen.cs, gen.cc
    # the product side
    import casclik_amd as pcc
    from casclik_amd import skills, sym as pcs
    from oracle import clik_oracle
    rng = np.random.default_rng(seed)
    B = 24
    worst = {"pinv": 0.0, "qp": 0.0, "qp_data": 0.0}
    counts = {"pinv_runs": 0, "qp_runs": 0, "mode_mismatch": 0, "status_mismatch": 0, "skipped": 0}
    for k in range(n_skills):
        robot = "iiwa" if rng.random() < 0.5 else "ur5"
        urdf, root, tip = gen.ROBOTS[robot]
        chain = gen.load_chain(urdf, root, tip)
        act = [j for j in chain if j["type"] != "fixed"]
        lower, upper = np.array([j["lower"] for j in act]), np.array([j["upper"] for j in act])
        vmax = np.array([j["velocity"] for j in act])
        n = len(act)
        rec = recipe(rng, n)
        fk = skills.iiwa() if robot == "iiwa" else skills.ur5()
        ref_spec = build(rec, rcs, rcc, gen.make_T_fk(chain), gen.ori_err, lower, upper, vmax)
        own_spec = build(rec, pcs, pcc, fk["T_fk"], pcs.orientation_error, lower, upper, vmax)
        assert [c.label for c in ref_spec.constraints] == [c.label for c in own_spec.constraints], "priority sort differs"
        texts = []
        for built in (ref_spec, own_spec):          # print_constraints(): the same text from both packages
            import contextlib
            import io
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                built.print_constraints()
            texts.append(buf.getvalue())
        assert texts[0] == texts[1], ("print_constraints() differs", texts)
        Q = rng.uniform(0.7 * lower, 0.7 * upper, size=(B, n))
        X = rng.uniform(0.0, 2.0, size=(B, 1)) if rec["virtual"] else None
        t0 = float(rng.uniform(0.0, 2.0))
        what = "%s %s%s" % (robot, [t["kind"] for t in rec["tasks"]], " +virtual" if rec["virtual"] else "")
        # ---- PseudoInverseController
        n_sets = sum(1 for c in own_spec.constraints if type(c).__name__ == "SetConstraint")
        try:
            ctrl = rcc.PseudoInverseController(skill_spec=ref_spec, options=dict(rec["options"]))
            ctrl.setup_problem_functions()
            ref_dz, ref_mode = np.zeros((B, n + (1 if rec["virtual"] else 0))), np.zeros(B, dtype=int)
            for b in range(B):
                res = ctrl.solve(t0, Q[b], X[b]) if rec["virtual"] else ctrl.solve(t0, Q[b])
                ref_dz[b, :n] = res[0].full().ravel()
                if rec["virtual"]:
                    ref_dz[b, n:] = res[1].full().ravel()
                ref_mode[b] = ctrl.current_mode
            kappa = np.zeros(B)
            dz, mode = clik_oracle.pinv_solve_batch(own_spec, dict(rec["options"]), t0, Q, X=X, cond_out=kappa)
            counts["pinv_runs"] += 1
            same_mode = mode == ref_mode
            counts["mode_mismatch"] += int((~same_mode).sum())
            err = np.abs(dz - ref_dz).max(axis=1) / (1.0 + np.abs(ref_dz).max(axis=1))
            e = float(err[same_mode].max()) if same_mode.any() else 0.0
            worst["pinv"] = max(worst["pinv"], e)
            # the stated rule (tests/tolerances.py): every instance against max(FLOOR, FACTOR u kappa)
            tol_b = rtol_from_cond(kappa)
            posed = same_mode & (tol_b < ILL_POSED)
            over = float((err / tol_b)[posed].max()) if posed.any() else 0.0
            worst["pinv_over"] = max(worst.get("pinv_over", 0.0), over)
            line = "pinv modes %s (%d sets) err %.1e (%.2f x tol) mismatching modes %d" % (
                np.bincount(ref_mode + 1).tolist(), n_sets, e, over, int((~same_mode).sum()))
        except Exception as exc:          # (skills the reference itself refuses: reported, not compared)
            counts["skipped"] += 1
            line = "pinv skipped (%s: %s)" % (type(exc).__name__, str(exc)[:60])
        # ---- ReactiveQPController
        try:
            qp = rcc.ReactiveQPController(skill_spec=ref_spec)
            qp.setup_problem_functions()
            qp.setup_solver()
            H, A, lbA, ubA = clik_oracle.qp_data_batch(own_spec, t0, Q, X)
            qkappa = np.ones(B)
            odq, odx, oslack, ostatus = clik_oracle.qp_solve_batch(own_spec, t0, Q, X=X, cond_out=qkappa)
            qtol_b = rtol_from_cond(qkappa)
            e_data = e_sol = 0.0
            bad_status = 0
            for b in range(B):
                vals = [t0, Q[b]] + ([X[b]] if rec["virtual"] else [])
                rA = qp.A_func(*vals).full()
                rl, ru = qp.Blb_func(*vals).full().ravel(), qp.Bub_func(*vals).full().ravel()
                clip = lambda v: np.clip(v, -1e9, 1e9)        # noqa: E731
                e_data = max(e_data, np.abs(A[b] - rA).max(), np.abs(clip(lbA[b]) - clip(rl)).max(),
                             np.abs(clip(ubA[b]) - clip(ru)).max())
                try:
                    res = qp.solve(t0, Q[b], X[b]) if rec["virtual"] else qp.solve(t0, Q[b])
                    r_status = 0
                except RuntimeError:
                    r_status = 2
                if (r_status == 2) != (ostatus[b] == 2):
                    # who is right?  an LP over the same rows (margin = how far the rows can be satisfied strictly)
                    from scipy.optimize import linprog
                    nv = A.shape[2]
                    Aub = np.vstack([np.hstack([A[b], np.ones((A.shape[1], 1))]), np.hstack([-A[b], np.ones((A.shape[1], 1))])])
                    bub = np.concatenate([np.clip(ubA[b], -1e9, 1e9), -np.clip(lbA[b], -1e9, 1e9)])
                    cost = np.zeros(nv + 1)
                    cost[-1] = -1.0
                    lp = linprog(cost, A_ub=Aub, b_ub=bub, bounds=[(None, None)] * nv + [(None, 1.0)], method="highs")
                    margin = lp.x[-1] if lp.status == 0 else float("nan")
                    print("      instance %d: reference status %d, oracle status %d; LP feasibility margin of the rows %.3e"
                          % (b, r_status, int(ostatus[b]), margin), flush=True)
                    if abs(margin) > 1e-7:
                        bad_status += 1
                    continue
                if r_status == 0 and ostatus[b] == 0:
                    rdq = res[0].full().ravel()
                    e_b = np.abs(odq[b] - rdq).max() / (1.0 + np.abs(rdq).max())
                    e_sol = max(e_sol, e_b)
                    if qtol_b[b] < ILL_POSED:
                        worst["qp_over"] = max(worst.get("qp_over", 0.0), e_b / qtol_b[b])
            # the initial problem (reactive_qp.py:300-459): virtual velocities and slack with the robot held still
            e_init = 0.0
            qp.setup_initial_problem_solver()
            for b in range(0, B, 6):
                kw0 = {"virtual_var0": X[b]} if rec["virtual"] else {}
                try:
                    r_virt, r_slack = qp.solve_initial_problem(t0, Q[b], **kw0)
                except RuntimeError:
                    continue
                o_virt, o_slack = clik_oracle.qp_initial_problem(own_spec, t0, Q[b], x0=X[b] if rec["virtual"] else None)
                for got, want in ((o_virt, r_virt), (o_slack, r_slack)):
                    if want is None or got is None:
                        assert (want is None) == (got is None), "initial problem: None pattern differs"
                        continue
                    w = np.asarray(want.full()).ravel()
                    e_init = max(e_init, float(np.abs(np.asarray(got).ravel() - w).max() / (1.0 + np.abs(w).max())))
            worst["qp_init"] = max(worst.get("qp_init", 0.0), e_init)
            counts["qp_runs"] += 1
            counts["status_mismatch"] += bad_status
            worst["qp"], worst["qp_data"] = max(worst["qp"], e_sol), max(worst["qp_data"], e_data)
            line += " | qp rows %d data err %.1e solution err %.1e infeasible %d status mismatches %d" % (
                A.shape[1], e_data, e_sol, int((ostatus == 2).sum()), bad_status)
        except Exception as exc:
            counts["skipped"] += 1
            line += " | qp skipped (%s: %s)" % (type(exc).__name__, str(exc)[:60])
        print("%3d %-60s %s" % (k, what[:60], line), flush=True)
    print("reference package over the stand-in vs the numpy oracle: %d random skills x %d instances; pinv runs %d, worst "
          "relative error %.2e, mismatching modes %d; QP runs %d, rows worst %.2e, minimiser worst %.2e, initial "
          "problem worst %.2e, status mismatches %d; refused by the reference or the oracle: %d"
          % (n_skills, B, counts["pinv_runs"], worst["pinv"], counts["mode_mismatch"], counts["qp_runs"],
             worst["qp_data"], worst["qp"], worst.get("qp_init", 0.0), counts["status_mismatch"], counts["skipped"]))
    print("against the stated rule (tests/tolerances.py, per instance max(FLOOR, FACTOR u kappa)): worst err / tol  pinv %.3f  "
          "qp %.3f   %s" % (worst.get("pinv_over", 0.0), worst.get("qp_over", 0.0),
                            "(all within)" if max(worst.get("pinv_over", 0.0), worst.get("qp_over", 0.0)) <= 1.0 else "<-- BEYOND THE RULE"))


if __name__ == "__main__":
    main()
