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
