"""Case tables shared by tests/golden/make_ref_golden.py (REFERENCE classes over the stand-in casadi) and the tests
(the product's classes): parametrised by the symbolic module and the class namespace, importing neither."""
import numpy as np


def tangent_cone_api_cases(cs, cc):
    """the controller's PUBLIC tangent-cone functions (pseudo_inverse.py:132-257) on three-joint toy skills: a
    time-dependent 1-D set with a virtual and an input variable, and a multidimensional set with a virtual variable
    (with an input_var the reference cannot build that one: :219-221 appends a string to the variable list).
    Per case (function, argument rows in the function's own order)."""
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 3), cs.MX.sym("dq", 3)
    x, dx, y = cs.MX.sym("x"), cs.MX.sym("dx"), cs.MX.sym("y", 2)
    one = cc.SetConstraint(label="one d", expression=q[0] * cs.sin(q[1]) + 0.1 * t - x + y[0], set_min=-0.2, set_max=0.3)
    box = cc.SetConstraint(label="box", expression=cs.vertcat(q[0] + 0.5 * q[2], q[1] * x, q[2] - 0.2 * t),
                           set_min=np.array([-0.3, -0.2, -0.4]), set_max=np.array([0.3, 0.2, 0.4]))
    task = cc.EqualityConstraint(label="task", expression=q - 0.1, constraint_type="soft")
    with_input = cc.SkillSpecification("toy", t, q, dq, virtual_var=x, virtual_vel_var=dx, input_var=y,
                                       constraints=[one, task])
    without = cc.SkillSpecification("toy", t, q, dq, virtual_var=x, virtual_vel_var=dx, constraints=[box, task])
    c1 = cc.PseudoInverseController(skill_spec=with_input)
    c2 = cc.PseudoInverseController(skill_spec=without, options={"multidim_sets": True})
    rng = np.random.default_rng(77)
    rows1, rows2 = [], []
    for k in range(160):
        scale = 0.15 if k % 4 == 0 else 0.6            # (a quarter of the points inside the sets)
        tt, qq, xx = float(rng.uniform(0, 2)), rng.uniform(-scale, scale, 3), float(rng.uniform(0.5, 1.5))
        yy, vq, vx = rng.uniform(-0.1, 0.1, 2), rng.normal(size=3), float(rng.normal())
        rows1.append([tt, qq, xx, yy, vq, vx])
        rows2.append([tt, qq, xx, vq, vx])
    return {"one": (c1.get_in_tangent_cone_function(one), rows1),
            "box": (c2.get_in_tangent_cone_function_multidim(box), rows2)}


def pinv_api_cases(cs, cc):
    """PseudoInverseController.pinv (pseudo_inverse.py:92-105) as a public method: [(name, J, result matrix)] for wide,
    square and tall constant matrices under the damped rule (default and a large damping factor) and under "standard" """
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 3)
    spec = cc.SkillSpecification("toy", t, q, constraints=[cc.EqualityConstraint(label="task", expression=q)])
    rng = np.random.default_rng(5)
    out = []
    for opt_name, options in (("damped", None), ("damped_1e-2", {"damping_factor": 1e-2}), ("standard", {"pinv_method": "standard"})):
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=options)
        for shape in ((2, 5), (4, 4), (6, 3)):
            if opt_name == "standard" and shape[0] > shape[1]:
                continue        # (cs.pinv of a tall matrix: the stand-in and CasADi agree only on the wide rule)
            J = rng.normal(size=shape)
            res = ctrl.pinv(cs.DM(J))
            val = res.full() if hasattr(res, "full") else res.toarray()
            out.append(("%s_%dx%d" % (opt_name, shape[0], shape[1]), J, np.asarray(val, dtype=float)))
    return out


def api_value_cases(cs, cc):
    """Values of the small public methods - BaseConstraint.size / jacobian / jtimes / nullspace, the constraint classes'
    defaults and class attributes, SkillSpecification's counters, flags, priority sort and re-sort on assignment
    (constraints.py:21-86, skill_specification.py:60-250) - on a toy skill, as one JSON-able dict."""
    def num(v):
        v = v.full() if hasattr(v, "full") else (v.toarray() if hasattr(v, "toarray") else v)
        return np.round(np.asarray(v, dtype=float), 10).tolist()
    t = cs.MX.sym("t"); q = cs.MX.sym("q", 3); dq = cs.MX.sym("dq", 3); x = cs.MX.sym("x"); dx = cs.MX.sym("dx"); y = cs.MX.sym("y", 2)
    out = {}
    e = cs.vertcat(q[0] * cs.sin(q[1]) + t * x, q[2] ** 2 - y[0] * q[0])
    c = cc.EqualityConstraint(label="a", expression=e, gain=2.0, constraint_type="soft", priority=3)
    pt = [0.3, np.array([0.2, -0.4, 0.7]), 1.5, np.array([0.5, -0.1]), np.array([1.0, 2.0, -1.0])]
    def F(expr): return cs.Function("f", [t, q, x, y, dq], [expr])(*pt)
    out["size"] = list(c.size())
    out["jac_q"] = num(F(c.jacobian(q))); out["jac_t"] = num(F(c.jacobian(t))); out["jac_x"] = num(F(c.jacobian(x))); out["jac_y"] = num(F(c.jacobian(y)))
    out["jtimes"] = num(F(c.jtimes(q, dq)))
    try: out["nullspace"] = num(F(c.nullspace(q)))
    except Exception as ex: out["nullspace"] = "EXC " + type(ex).__name__
    s1 = cc.SetConstraint(label="s1", expression=q[0], set_min=-1.0, set_max=1.0, priority=5, constraint_type="soft")
    s2 = cc.SetConstraint(label="s2", expression=q[1], priority=1)
    ve = cc.VelocityEqualityConstraint(label="ve", expression=q[2], target=0.2, priority=2, constraint_type="soft")
    vs = cc.VelocitySetConstraint(label="vs", expression=q, priority=2)
    out["defaults"] = [num(s2.set_min), num(s2.set_max), num(vs.set_min), num(vs.set_max), s2.gain, ve.gain, vs.gain, s2.constraint_type, ve.slack_weight]
    spec = cc.SkillSpecification("s", t, q, dq, virtual_var=x, virtual_vel_var=dx, input_var=y, constraints=[c, s1, s2, ve, vs])
    def snap(spec): return {"order": [k.label for k in spec.constraints], "n": [spec.n_robot_var, spec.n_virtual_var, spec.n_input_var, spec.n_slack_var], "has": [bool(spec._has_virtual), bool(spec._has_input)], "count": dict(spec.count_constraints())}
    out["spec"] = snap(spec)
    spec.constraints = [vs, ve, s2, s1]
    out["spec_after_set"] = snap(spec)
    spec2 = cc.SkillSpecification("s", t, q, dq, constraints=[cc.EqualityConstraint(label="only", expression=q)])
    out["spec2"] = snap(spec2)
    out["attrs"] = {k: getattr(spec2, k) for k in ("label",)}
    for name, obj in (("eq", c), ("set", s1), ("veq", ve), ("vset", vs)):
        out["cls_" + name] = [obj.constraint_class, obj.constraint_type, obj.priority, obj.label]
    # controllers at construction: the mode table for 0 ... 4 SetConstraints (pseudo_inverse.py:107-130), counters, option
    # defaults of both controllers (pseudo_inverse.py:42-66, reactive_qp.py:141-173)
    q5 = cs.MX.sym("q", 5)
    for n in range(0, 5):
        cons = [cc.EqualityConstraint(label="e", expression=q5[0])] + [
            cc.SetConstraint(label="s%d" % i, expression=q5[i], set_min=-1.0, set_max=1.0, priority=i) for i in range(n)]
        ctrl = cc.PseudoInverseController(skill_spec=cc.SkillSpecification("s", t, q5, constraints=cons))
        out["pinv_ctrl_%d_sets" % n] = {
            "map": [list(map(int, m)) for m in ctrl.activation_map], "n_modes": ctrl.n_modes,
            "n_set": ctrl.n_set_constraints, "n_state": ctrl.n_state_var,
            "opts": {k: ctrl.options[k] for k in ("pinv_method", "damping_factor", "feedforward", "multidim_sets",
                                                  "converge_final_set_to_max")},
            "optkeys": sorted(ctrl.options.keys())}
    soft = cc.SkillSpecification("s", t, q5, constraints=[cc.EqualityConstraint(label="e", expression=q5[0],
                                                                                constraint_type="soft")])
    qp = cc.ReactiveQPController(skill_spec=soft)
    out["qp_ctrl"] = {"optkeys": sorted(qp.options.keys()), "solver_name": qp.options["solver_name"],
                      "solver_opts": {k: qp.options["solver_opts"][k] for k in sorted(qp.options["solver_opts"])},
                      "function_opts": {k: qp.options["function_opts"][k] for k in sorted(qp.options["function_opts"])},
                      "mu": qp.weight_shifter, "type": qp.controller_type}
    return out
