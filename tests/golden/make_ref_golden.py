#!/usr/bin/env python3
"""Generate tests/golden/ref_pins.npz by running the REFERENCE's own Python (casclik from
/root/reference) over the stand-in casadi module (tests/golden/refshim).  Build container only:
the reference does not exist on the GPU box, the fixture (inputs + outputs, data only) travels.

    python tests/golden/make_ref_golden.py

What this pins and what it does not (see refshim/casadi/__init__.py): the reference's control flow and
formulas are the reference's (its files are imported, not restated); the arithmetic under them is numpy,
not CasADi.  Forward kinematics: the reference gets T_fk from urdf2casadi (not vendored); here the chain
is read from the reference's own URDF files and multiplied out with cs.sin / cs.cos / cs.mtimes.
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import casadi as cs          # noqa: E402  (the stand-in)
import casclik as cc         # noqa: E402  (the reference)
import pin_skills            # noqa: E402

assert cs.__file__.startswith(HERE) and cc.__file__.startswith(REF), (cs.__file__, cc.__file__)


def rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def load_chain(urdf, root, tip):
    """joints root -> tip of a URDF: (type, origin R, origin p, axis, lower, upper, velocity)"""
    tree = ET.parse(urdf).getroot()
    by_child = {j.find("child").attrib["link"]: j for j in tree.findall("joint")}
    chain, link = [], tip
    while link != root:
        j = by_child[link]
        org = j.find("origin")
        xyz = [float(v) for v in (org.attrib.get("xyz", "0 0 0") if org is not None else "0 0 0").split()]
        rpy = [float(v) for v in (org.attrib.get("rpy", "0 0 0") if org is not None else "0 0 0").split()]
        ax = j.find("axis")
        axis = [float(v) for v in (ax.attrib["xyz"] if ax is not None else "1 0 0").split()]
        lim = j.find("limit")
        chain.append(dict(type=j.attrib["type"], R=rpy_matrix(*rpy), p=np.array(xyz), axis=np.array(axis),
                          lower=float(lim.attrib.get("lower", 0)) if lim is not None else 0.0,
                          upper=float(lim.attrib.get("upper", 0)) if lim is not None else 0.0,
                          velocity=float(lim.attrib.get("velocity", 0)) if lim is not None else 0.0))
        link = j.find("parent").attrib["link"]
    return chain[::-1]


def axis_rotation(axis, c, s):
    """Rodrigues rotation about a unit axis, entries as expressions of c = cos q, s = sin q"""
    x, y, z = axis
    C = 1 - c
    return [[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
            [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
            [z * x * C - y * s, z * y * C + x * s, c + z * z * C]]


def make_T_fk(chain):
    def T_fk(q):
        T = cs.DM.eye(4)
        k = 0
        for j in chain:
            O = np.eye(4)
            O[:3, :3], O[:3, 3] = j["R"], j["p"]
            T = cs.mtimes(T, cs.DM(O))
            if j["type"] in ("revolute", "continuous"):
                rows = axis_rotation(j["axis"], cs.cos(q[k]), cs.sin(q[k]))
                Rq = cs.vertcat(*[cs.horzcat(*(list(r) + [0.0])) for r in rows], cs.horzcat(0.0, 0.0, 0.0, 1.0))
                T = cs.mtimes(T, Rq)
                k += 1
            elif j["type"] == "prismatic":
                Pq = cs.vertcat(*[cs.horzcat(*(list(np.eye(3)[i]) + [j["axis"][i] * q[k]])) for i in range(3)],
                                cs.horzcat(0.0, 0.0, 0.0, 1.0))
                T = cs.mtimes(T, Pq)
                k += 1
        return T
    return T_fk


def make_dh_T_fk(link_lengths, link_twists, link_offsets):
    """T = prod_i Rot_z(q_i) Trans_z(d_i) Trans_x(a_i) Rot_x(alpha_i): the classic Denavit-Hartenberg table
    ur5_moe2016_example2.ipynb cell 2 hands to urdf2casadi's `from_denavit_hartenberg`, multiplied out with the stand-in"""
    def T_fk(q):
        T = cs.DM.eye(4)
        for i, (a, alpha, d) in enumerate(zip(link_lengths, link_twists, link_offsets)):
            c, s = cs.cos(q[i]), cs.sin(q[i])
            Rz = cs.vertcat(cs.horzcat(c, -s, 0.0, 0.0), cs.horzcat(s, c, 0.0, 0.0), cs.horzcat(0.0, 0.0, 1.0, 0.0),
                            cs.horzcat(0.0, 0.0, 0.0, 1.0))
            ca, sa = np.cos(alpha), np.sin(alpha)
            link = np.array([[1.0, 0.0, 0.0, a], [0.0, ca, -sa, 0.0], [0.0, sa, ca, d], [0.0, 0.0, 0.0, 1.0]])
            T = cs.mtimes(T, cs.mtimes(Rz, cs.DM(link)))
        return T
    return T_fk


def quat_to_rot(qt):
    x, y, z, w = qt[0], qt[1], qt[2], qt[3]
    return [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]


def ori_err(R, quat):
    """e_o = 1/2 sum_c r_c x rd_c  (SURVEY.md 8(d)), columns of R and of the target rotation"""
    Rd = quat_to_rot(quat)
    acc = None
    for c in range(3):
        r = [R[i, c] for i in range(3)]
        d = [Rd[i][c] for i in range(3)]
        cr = [r[1] * d[2] - r[2] * d[1], r[2] * d[0] - r[0] * d[2], r[0] * d[1] - r[1] * d[0]]
        acc = cr if acc is None else [a + b for a, b in zip(acc, cr)]
    return cs.vertcat(*[0.5 * a for a in acc])


def quat_from_matrix(R):
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
    out = np.zeros(4)
    out[i], out[j], out[k], out[3] = 0.25 * s, (R[j, i] + R[i, j]) / s, (R[k, i] + R[i, k]) / s, (R[k, j] - R[j, k]) / s
    return out


def numeric_fk(T_fk, n, qv):
    q = cs.MX.sym("q", n)
    return cs.Function("fk", [q], [T_fk(q)])(qv).full()


ROBOTS = {
    "iiwa": (os.path.join(REF, "examples/notebooks/urdf/lbr_iiwa_14_r820.urdf"), "base_link", "tool0"),
    "ur5": (os.path.join(REF, "examples/notebooks/urdf/ur5.urdf"), "base_link", "tool0"),
}
PLAN = [   # (fixture name, robot, case, B, input distribution, times)
    ("iiwa_position", "iiwa", "position", 64, "interior", [0.0]),
    ("iiwa_position_standard", "iiwa", "position_standard", 48, "interior", [0.0]),
    ("iiwa_pose", "iiwa", "pose", 64, "interior", [0.0]),
    ("ur5_pose", "ur5", "pose", 48, "interior", [0.0]),
    ("iiwa_stack_const", "iiwa", "stack_const", 96, "mixed", [0.0]),
    ("ur5_stack_const", "ur5", "stack_const", 64, "mixed", [0.0]),
    ("iiwa_stack_const_time", "iiwa", "stack_const_time", 48, "mixed", [0.3, 1.7]),
    ("iiwa_stack_sets1d", "iiwa", "stack_sets1d", 96, "mixed", [0.0]),
    ("iiwa_stack_noff", "iiwa", "stack_noff", 48, "mixed", [0.9]),
    ("iiwa_conv_last", "iiwa", "conv_last", 64, "mixed", [0.0]),
    ("iiwa_veleq_first", "iiwa", "veleq_first", 48, "interior", [0.0]),
    ("iiwa_qp_pose", "iiwa", "qp_pose", 64, "interior", [0.0]),
    ("ur5_qp_limits", "ur5", "qp_limits", 64, "mixed", [0.0]),
    ("iiwa_qp_path", "iiwa", "qp_path", 48, "interior", [0.0]),
    # (appended: the seeds of the entries above follow their position)
    ("iiwa_sym_attrs", "iiwa", "sym_attrs", 64, "mixed", [0.4, 2.3]),
    ("ur5_qp_sym_attrs", "ur5", "qp_sym_attrs", 48, "mixed", [1.1]),
    # round 3
    ("iiwa_stack_boundary", "iiwa", "stack_boundary", 192, "boundary", [0.0]),
    ("ur5_qp_wall", "ur5", "qp_wall", 96, "mixed", [0.0]),
    ("iiwa_two_frames", "iiwa", "two_frames", 64, "interior", [0.0]),
    ("iiwa_qp_two_virtual", "iiwa", "qp_two_virtual", 48, "interior", [0.0]),
    ("iiwa_sets_per_joint", "iiwa", "sets_per_joint", 160, "narrow", [0.0]),
    # round 4: atan2 / asin / acos / atan / tanh / fmin / fmax in task errors
    ("iiwa_heading", "iiwa", "heading", 64, "mixed", [0.0, 1.3]),
    ("ur5_qp_heading", "ur5", "qp_heading", 48, "interior", [0.7]),
]
# offsets from a joint limit the "boundary" distribution plants (pseudo_inverse.py:222-252 thresholds e - bound
# at 1e-12): exactly on the limit, either side of the 1e-12 margin, and up to 1e-6 away
BOUNDARY_OFFSETS = [0.0, 5e-13, -5e-13, 1e-12, -1e-12, 2e-12, -2e-12, 1e-10, -1e-10, 1e-9, -1e-9, 1e-7, -1e-7,
                    1e-6, -1e-6]


def inputs(chain, T_fk, B, dist, seed):
    act = [j for j in chain if j["type"] != "fixed"]
    lo, hi = np.array([j["lower"] for j in act]), np.array([j["upper"] for j in act])
    rng = np.random.default_rng(seed)
    if dist == "interior":
        Q = rng.uniform(0.9 * lo, 0.9 * hi, size=(B, len(lo)))
    elif dist == "narrow":
        # around sets drawn at 0.3 of the joint ranges (pin_skills.sets_per_joint): each joint outside with p = 1/6
        Q = rng.uniform(0.36 * lo, 0.36 * hi, size=(B, len(lo)))
    else:
        r = hi - lo
        Q = rng.uniform(lo - 0.05 * r, hi + 0.05 * r, size=(B, len(lo)))
    if dist == "boundary":
        # interior configurations with one to three joints planted within 1e-6 of a limit (either limit, either side)
        Q = rng.uniform(0.9 * lo, 0.9 * hi, size=(B, len(lo)))
        for b in range(B):
            for j in rng.choice(len(lo), size=int(rng.integers(1, 4)), replace=False):
                lim = lo[j] if rng.random() < 0.5 else hi[j]
                Q[b, j] = lim + BOUNDARY_OFFSETS[int(rng.integers(len(BOUNDARY_OFFSETS)))]
    Qd = rng.uniform(0.8 * lo, 0.8 * hi, size=(B, len(lo)))
    Y = np.zeros((B, 7))
    for b in range(B):
        T = numeric_fk(T_fk, len(lo), Qd[b])
        Y[b, :3], Y[b, 3:] = T[:3, 3], quat_from_matrix(T[:3, :3])
    return Q, Y


def check_reference_held_outputs():
    """Tie the stand-in casadi to the outputs the reference itself stores, through the
    REFERENCE package:
      * print_constraints() of the notebooks' skills, rebuilt with reference classes over the stand-in, must equal
        the text the notebooks store verbatim (tests/golden/notebook_prints.json): pins MX.nnz() of a Jacobian as the
        dependence test (skill_specification.py:228-250 -> _has_virtual / _has_input), the stable priority sort
        (:139-152) and the per-class counters;
      * forward kinematics multiplied out with the stand-in's sin / cos / mtimes from the reference's UR5 URDF must
        give ||p|| = 1.0192 at UR5_home and the home dual quaternion the notebooks print."""
    import io
    import json
    data = json.load(open(os.path.join(HERE, "notebook_prints.json")))
    for fx in data["prints"]:
        t, p, dp = cs.MX.sym("t"), cs.MX.sym("p"), cs.MX.sym("dp")
        x, dx = (cs.MX.sym("x"), cs.MX.sym("dx")) if fx["virtual"] else (None, None)
        sym = {"q": p, "x": x, "t": t}
        cons = []
        for c in fx["constraints"]:
            expr = 0.5
            for d in c["depends"]:
                expr = expr + (0.4 * cs.sin(0.3 * sym[d]) if d != "q" else sym[d])
            kw = dict(label=c["label"], expression=expr, priority=c["priority"], constraint_type=c["constraint_type"])
            if c["cls"] == "SetConstraint":
                cons.append(cc.SetConstraint(set_min=0.0, set_max=1.0, **kw))
            elif c["cls"] == "VelocitySetConstraint":
                cons.append(cc.VelocitySetConstraint(set_min=-0.275, set_max=0.275, **kw))
            else:
                cons.append(cc.EqualityConstraint(gain=1.0, **kw))
        spec = cc.SkillSpecification(label=fx["label"], time_var=t, robot_var=p, robot_vel_var=dp, virtual_var=x,
                                     virtual_vel_var=dx, constraints=cons)
        buf, old = io.StringIO(), sys.stdout
        sys.stdout = buf
        try:
            spec.print_constraints()
        finally:
            sys.stdout = old
        assert buf.getvalue() == fx["stdout"], (fx["notebook"], fx["cell"], buf.getvalue(), fx["stdout"])
    urdf, root, tip = ROBOTS["ur5"]
    T_fk = make_T_fk(load_chain(urdf, root, tip))
    home = [0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0]
    T = numeric_fk(T_fk, 6, home)
    assert abs(np.linalg.norm(T[:3, 3]) - 1.0192) < 5e-5, np.linalg.norm(T[:3, 3])
    assert np.allclose(T[:3, 3], [0.0, 0.19145, 1.001059], atol=1e-9)
    assert np.allclose(T[:3, :3], [[1, 0, 0], [0, 0, 1], [0, -1, 0]], atol=1e-9)
    # the dual quaternion [r; 1/2 t (x) r] of that frame (ur5_dual_quaternion_comparison_of_controllers.ipynb cell 7)
    r = quat_from_matrix(T[:3, :3])
    tx, ty, tz = T[:3, 3]
    x_, y_, z_, w_ = r
    d = 0.5 * np.array([tx * w_ + ty * z_ - tz * y_, -tx * z_ + ty * w_ + tz * x_, tx * y_ - ty * x_ + tz * w_,
                        -tx * x_ - ty * y_ - tz * z_])
    printed = np.array([-0.707107, -3.46237e-12, -3.46237e-12, 0.707107, -3.40946e-13, -0.28624, 0.421616, 3.21923e-13])
    Qh = np.concatenate([r, d])
    assert min(np.abs(Qh - printed).max(), np.abs(Qh + printed).max()) < 5e-7, Qh
    print("reference-held outputs reproduced through the reference package over the stand-in: %d print_constraints() "
          "texts verbatim, UR5 home ||p|| = %.5f, home dual quaternion" % (len(data["prints"]), np.linalg.norm(T[:3, 3])))
    return {"n_prints": len(data["prints"]), "ur5_home_norm": float(np.linalg.norm(T[:3, 3]))}


def check_against_the_stored_figures():
    """The strongest tie of the stand-in casadi to the REAL one: the closed-loop figures the reference's notebooks store
    (their author's runs on CasADi + qpOASES), digitised by make_figure_pins.py.  The REFERENCE controllers over the
    stand-in run the same notebook loops here - the cart notebook's six runs (QP and pinv: point, trajectory leaving
    the rail, path with a virtual variable) and the double pendulum's two QP runs (task-space SetConstraints) - and
    every simulated curve has to pass within a pixel of every digitised sample.  Returns the worst deviation [px]."""
    import figure_skills as fs
    worst_all, n_curves = 0.0, 0
    for case in fs.CASES:
        kind, spec, dt, p0, virt = fs.build(case, cs, cc)
        if kind == "qp":
            ctrl = cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0])
        else:
            ctrl = cc.PseudoInverseController(skill_spec=spec)
        ctrl.setup_problem_functions()
        ctrl.setup_solver()

        def solve(t, p, x, ctrl=ctrl, virt=virt):
            res = ctrl.solve(t, p, x) if virt else ctrl.solve(t, p)
            return float(res[0].toarray()[0, 0]), (float(res[1].toarray()[0, 0]) if virt else None)
        t_sim, p_sim, dp_sim = fs.simulate(case, solve)
        for curve in (["dp"] if case.endswith("point") else ["p", "dp"]):
            worst, n = fs.deviation_in_pixels(case, curve, t_sim, p_sim if curve == "p" else dp_sim)
            assert worst < 1.0, ("reference + stand-in misses the stored figure", case, curve, worst)
            worst_all, n_curves = max(worst_all, worst), n_curves + 1
    for case in fs.PENDULUM_CASES:
        ctrl = cc.ReactiveQPController(skill_spec=fs.pendulum_skill(case, cs, cc), robot_var_weights=[1.0, 1.0])
        ctrl.setup_problem_functions()
        ctrl.setup_solver()
        t_sim, _, dq_sim, p_sim = fs.simulate_pendulum(case, lambda t, q, ctrl=ctrl: ctrl.solve(t, q)[0].toarray()[:, 0])
        curves = ([("pend_point_dq", "dq0", dq_sim[:, 0]), ("pend_point_dq", "dq1", dq_sim[:, 1]),
                   ("pend_point_p", "px", p_sim[:, 0]), ("pend_point_p", "py", p_sim[:, 1])] if case == "pend_point"
                  else [("pend_track_p", "px", p_sim[:, 0]), ("pend_track_p", "py", p_sim[:, 1])])
        for fig, curve, values in curves:
            worst, n = fs.deviation_in_pixels(fig, curve, t_sim, values)
            assert worst < 1.0, ("reference + stand-in misses the stored figure", fig, curve, worst)
            worst_all, n_curves = max(worst_all, worst), n_curves + 1
    print("reference package + stand-in casadi: 9 stored closed-loop figures (%d curves) reproduced, worst deviation "
          "%.2f pixel rows" % (n_curves, worst_all))
    # ur5_moe2016_example2.ipynb cells 2-12: the REFERENCE's PseudoInverseController (8 modes; multidim_sets with the box
    # active for 45 % of the run) and ReactiveQPController over the stand-in, 10000 ticks each, against cells 13-27's
    # figures (interval pins: tests/golden/moe_figure_pins.py)
    T_fk = make_dh_T_fk(**fs.MOE_DH)
    qsym = cs.MX.sym("q", 6)
    p_num = cs.Function("p", [qsym], [T_fk(qsym)[:3, 3]])
    worst_moe, n_pins = 0.0, 0
    for case in fs.MOE_CASES:
        kind, sit = case.split("_")
        spec, _, _ = fs.moe_skill(sit, cs, cc, T_fk)
        if kind == "pinv":
            ctrl = cc.PseudoInverseController(skill_spec=spec, options={"multidim_sets": True} if sit == "multidim" else None)
        else:
            ctrl = cc.ReactiveQPController(skill_spec=spec)
        ctrl.setup_problem_functions()
        ctrl.setup_solver()
        ctrl.setup_initial_problem_solver()
        state = {"slack": ctrl.solve_initial_problem(0, fs.MOE_HOME)[-1]}

        def solve(t, q, ctrl=ctrl, kind=kind, state=state):
            res = ctrl.solve(t, q, warmstart_slack_var=state["slack"])
            if res[-1] is not None:
                state["slack"] = res[-1].toarray()[:, 0]
            return res[0].toarray()[:, 0], (ctrl.current_mode if kind == "pinv" else None)
        t_sim, q_sim, p_sim, e_sim, mode_sim = fs.simulate_moe(solve, lambda q: p_num(q).full()[:, 0])
        for key, worst, n, where in fs.moe_pins(case, t_sim, p_sim, e_sim, mode_sim):
            assert n >= 2 and worst < 1.0, ("reference + stand-in misses the stored figure", case, key, worst, where)
            worst_moe, n_pins = max(worst_moe, worst), n_pins + 1
        if case == "pinv_singular":
            assert fs.fill_deviation("moe_modes_separate", t_sim, mode_sim)[0] == 0
        print("   %s: %d ticks through the reference's solve()" % (case, len(t_sim) - 1))
    print("reference package + stand-in casadi: the Moe-2016 notebook's 12 stored figures (%d pins over 4 runs) reproduced, worst "
          "deviation %.2f pixel rows" % (n_pins, worst_moe))
    return max(worst_all, worst_moe)


def record_error_behaviour():
    """what the reference raises for the constructions of error_cases.py: class and message digest per case"""
    import json
    import error_cases
    out = {name: [kind, error_cases.digest(text)] for name, (kind, text) in error_cases.cases(cs, cc).items()}
    path = os.path.join(HERE, "ref_errors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, len(out), "cases,", sum(1 for v in out.values() if v[0] != "ok"), "of them raise")


def record_tangent_cone_api(out):
    import make_ref_golden_cases
    for name, (fn, rows) in make_ref_golden_cases.tangent_cone_api_cases(cs, cc).items():
        vals = [float(np.asarray(fn(*row).full()).ravel()[0]) for row in rows]
        out["tcapi_" + name] = np.array(vals)
        print("tangent-cone function %-4s in the cone for %d of %d argument rows" % (name, int(sum(vals)), len(vals)))


def record_pinv_api(out):
    import make_ref_golden_cases
    for name, J, val in make_ref_golden_cases.pinv_api_cases(cs, cc):
        out["pinvapi_" + name] = val
    print("pinv(): %d matrices recorded" % len([k for k in out if k.startswith("pinvapi_")]))


def return_pattern(res):
    """shape of what solve() returns (pseudo_inverse.py:512-556, reactive_qp.py:461-528): per tuple entry -1 for None,
    else rows * 100 + columns of its .toarray()"""
    return np.array([-1 if r is None else int(r.full().shape[0]) * 100 + int(r.full().shape[1]) for r in res])


def record_api_values():
    import json
    import make_ref_golden_cases
    path = os.path.join(HERE, "ref_api_values.json")
    with open(path, "w") as f:
        json.dump(make_ref_golden_cases.api_value_cases(cs, cc), f, indent=1, sort_keys=True)
    print("wrote", path)


def main():
    held = check_reference_held_outputs()
    held["figures_worst_px"] = check_against_the_stored_figures()
    record_error_behaviour()
    record_api_values()
    out = {"refheld_n_prints": np.array(held["n_prints"]), "refheld_ur5_home_norm": np.array(held["ur5_home_norm"]),
           "refheld_figures_worst_px": np.array(held["figures_worst_px"])}
    record_tangent_cone_api(out)
    record_pinv_api(out)
    for k, (name, robot, case, B, dist, times) in enumerate(PLAN):
        urdf, root, tip = ROBOTS[robot]
        chain = load_chain(urdf, root, tip)
        act = [j for j in chain if j["type"] != "fixed"]
        T_fk = make_T_fk(chain)
        n = len(act)
        lower, upper = [j["lower"] for j in act], [j["upper"] for j in act]
        vmax = [j["velocity"] for j in act]
        Q, Y = inputs(chain, T_fk, B, dist, seed=100 + k)
        # constant target of the input-free skills: the pose at a fixed interior configuration
        q_c = 0.35 * np.array(upper) * np.array([1, -1, 1, -1, 1, -1, 1][:n])
        Tc = numeric_fk(T_fk, n, q_c)
        consts = {"p_des": Tc[:3, 3], "quat_des": quat_from_matrix(Tc[:3, :3])}
        other = "ur5" if robot == "iiwa" else "iiwa"
        T_fk_alt = make_T_fk(load_chain(*ROBOTS[other]))
        env = pin_skills.Env(cs, cc, T_fk, ori_err, lower, upper, vmax, consts, T_fk_alt=T_fk_alt)
        built = pin_skills.CASES[case](env)
        spec, ny, nx = built["spec"], built["ny"], built.get("nx", 0)
        rng = np.random.default_rng(900 + k)
        X = rng.uniform(-0.2, 1.2, size=(B, nx)) if nx else None
        out[name + "_Q"], out[name + "_t"] = Q, np.array(times)
        out[name + "_p_des"], out[name + "_quat_des"] = consts["p_des"], consts["quat_des"]
        if ny:
            out[name + "_Y"] = Y[:, :ny]
        if nx:
            out[name + "_X"] = X
        if built["controller"] == "pinv":
            ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(built["options"]))
            ctrl.setup_problem_functions()
            dq = np.zeros((len(times), B, n))
            mode = np.zeros((len(times), B), dtype=np.int32)
            for ti, t in enumerate(times):
                for b in range(B):
                    kw = {"input_var": Y[b, :ny]} if ny else {}
                    res = ctrl.solve(t, Q[b], **kw)
                    dq[ti, b] = res[0].full().reshape(-1)
                    mode[ti, b] = ctrl.current_mode
            out[name + "_dq"], out[name + "_mode"] = dq, mode
            out[name + "_ret"] = return_pattern(res)
            print("%-26s modes %s  max|dq| %.3g  (%d constraints, order %s)" % (
                name, np.bincount(mode.reshape(-1) + 1), np.abs(dq).max(), len(spec.constraints),
                [c.label for c in spec.constraints]))
        else:
            ctrl = cc.ReactiveQPController(skill_spec=spec, options=dict(built["options"]))
            ctrl.setup_problem_functions()
            ctrl.setup_solver()
            ctrl.setup_initial_problem_solver()
            nv = n + nx + spec.n_slack_var
            nc = sum(c.expression.size()[0] for c in spec.constraints)
            H, A = np.zeros((B, nv)), np.zeros((B, nc, nv))
            lbA, ubA = np.zeros((B, nc)), np.zeros((B, nc))
            dq, dx, slack = np.zeros((B, n)), np.zeros((B, nx)), np.zeros((B, spec.n_slack_var))
            ivirt, islack = np.zeros((B, nx)), np.zeros((B, spec.n_slack_var))
            status = np.zeros(B, dtype=np.int32)
            worst = 0.0
            for b in range(B):
                kw = {}
                if ny:
                    kw["input_var"] = Y[b, :ny]
                if nx:
                    kw["virtual_var"] = X[b]
                vals_ = [times[0], Q[b]] + ([X[b]] if nx else []) + ([Y[b, :ny]] if ny else [])
                try:
                    rq, rx, rs = ctrl.solve(times[0], Q[b], **kw)
                    out[name + "_ret"] = return_pattern((rq, rx, rs))
                except RuntimeError:
                    # the reference surfaces an infeasible QP as the solver's RuntimeError (reactive_qp.py:491-513):
                    # recorded as status 2 (rows of the data functions are still the reference's)
                    status[b] = 2
                    Hb = ctrl.H_func(*vals_).full()
                    H[b], A[b] = np.diag(Hb), ctrl.A_func(*vals_).full()
                    lbA[b], ubA[b] = ctrl.Blb_func(*vals_).full().reshape(-1), ctrl.Bub_func(*vals_).full().reshape(-1)
                    # ... and only if the rows really admit no point (an LP says so; a stand-in solver that merely
                    # failed to converge must not become a pinned "infeasible")
                    from scipy.optimize import linprog
                    Aub = np.vstack([A[b], -A[b], ])
                    bub = np.concatenate([ubA[b], -lbA[b]])
                    lp = linprog(np.zeros(nv), A_ub=Aub, b_ub=bub, bounds=[(None, None)] * nv, method="highs")
                    assert lp.status == 2, ("stand-in QP failed on a feasible problem", name, b, lp.status)
                    dq[b], slack[b] = np.nan, np.nan
                    continue
                vals = [times[0], Q[b]] + ([X[b]] if nx else []) + ([Y[b, :ny]] if ny else [])
                Hb = ctrl.H_func(*vals).full()
                H[b], A[b] = np.diag(Hb), ctrl.A_func(*vals).full()
                lbA[b], ubA[b] = ctrl.Blb_func(*vals).full().reshape(-1), ctrl.Bub_func(*vals).full().reshape(-1)
                dq[b] = rq.full().reshape(-1)
                if nx:
                    dx[b] = rx.full().reshape(-1)
                slack[b] = rs.full().reshape(-1)
                # the stand-in QP solver's answer must satisfy the KKT conditions of the reference's data
                x = ctrl.res["x"].full().reshape(-1)
                lam = ctrl.res["lam_a"].full().reshape(-1)
                ax = A[b] @ x
                stat = np.abs(Hb @ x + A[b].T @ lam).max()
                feas = max(0.0, (ax - ubA[b]).max(), (lbA[b] - ax).max())
                comp = max(np.abs(np.where(lam > 0, lam * (ax - ubA[b]), 0.0)).max(),
                           np.abs(np.where(lam < 0, lam * (ax - lbA[b]), 0.0)).max())
                worst = max(worst, stat, feas, comp)
                iv, isl = ctrl.solve_initial_problem(times[0], Q[b], virtual_var0=(X[b] if nx else None),
                                                     input_var0=(Y[b, :ny] if ny else None))
                if nx:
                    ivirt[b] = iv.full().reshape(-1)
                islack[b] = isl.full().reshape(-1)
            assert worst < 1e-8, worst
            out.update({name + "_H": H, name + "_A": A, name + "_lbA": lbA, name + "_ubA": ubA, name + "_dq": dq,
                        name + "_dx": dx, name + "_slack": slack, name + "_init_virt": ivirt,
                        name + "_init_slack": islack})
            if status.any():
                out[name + "_status"] = status
            print("%-26s nv %d nc %d  KKT residual of the stand-in QP %.2e  max|dq| %.3g  infeasible %d" % (
                name, nv, nc, worst, np.nanmax(np.abs(dq)), int((status == 2).sum())))
    np.savez_compressed(os.path.join(HERE, "ref_pins.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_pins.npz"), "%d arrays" % len(out))


if __name__ == "__main__":
    main()
