"""Constructions the reference refuses (or accepts), as one table for both sides: tests/golden/make_ref_golden.py runs it
with the REFERENCE classes over the stand-in casadi and stores the exception class and a digest of the message per
case (tests/golden/ref_errors.json - data, not text); tests/test_api.py runs it with the product's classes and compares.
Raise sites covered: constraints.py:55-64, :124, :129-131, :209, :240, :266, :274-276; skill_specification.py:79-81,
:111-113; reactive_qp.py:71-77, :95-101, :124-130.  Imports nothing of either side."""
import hashlib

import numpy as np


def digest(message):
    return hashlib.sha1(message.encode("utf-8")).hexdigest()[:16]


def cases(cs, cc):
    t = cs.MX.sym("t"); q = cs.MX.sym("q", 3); dq = cs.MX.sym("dq", 3); x = cs.MX.sym("x"); dx = cs.MX.sym("dx")
    out = {}
    def run(name, f):
        try:
            f(); out[name] = ("ok", "")
        except Exception as e:
            out[name] = (type(e).__name__, str(e))
    run("eq_gain_str", lambda: cc.EqualityConstraint(label="a", expression=q, gain="fast"))
    run("eq_gain_dims", lambda: cc.EqualityConstraint(label="a", expression=q, gain=[1.0, 2.0]))
    run("eq_gain_list_ok", lambda: cc.EqualityConstraint(label="a", expression=q, gain=[1.0, 2.0, 3.0]))
    run("eq_gain_matrix_ok", lambda: cc.EqualityConstraint(label="a", expression=q, gain=np.eye(3)))
    run("eq_gain_matrix_dims", lambda: cc.EqualityConstraint(label="a", expression=q, gain=np.eye(2)))
    def add_prio():
        a = cc.EqualityConstraint(label="a", expression=q, priority=1); b = cc.EqualityConstraint(label="b", expression=q, priority=2); a + b
    run("eq_add_priority", add_prio)
    def add_type():
        a = cc.EqualityConstraint(label="a", expression=q, constraint_type="hard"); b = cc.EqualityConstraint(label="b", expression=q, constraint_type="soft"); a + b
    run("eq_add_type", add_type)
    def add_ok():
        a = cc.EqualityConstraint(label="a", expression=q); b = cc.EqualityConstraint(label="b", expression=q[0]); c = a + b; assert c.expression.size()[0] == 4
    run("eq_add_ok", add_ok)
    run("set_min_str", lambda: cc.SetConstraint(label="s", expression=q, set_min="low", set_max=[1, 1, 1]))
    run("set_max_str", lambda: cc.SetConstraint(label="s", expression=q, set_min=[0, 0, 0], set_max="hi"))
    run("set_min_list", lambda: cc.SetConstraint(label="s", expression=q, set_min=[0, 0, 0], set_max=np.array([1.0, 1.0, 1.0])))
    run("set_ndarray_ok", lambda: cc.SetConstraint(label="s", expression=q, set_min=np.zeros(3), set_max=np.ones(3)))
    run("set_default_bounds_ok", lambda: cc.SetConstraint(label="s", expression=q[0]))
    run("set_dims", lambda: cc.SetConstraint(label="s", expression=q, set_min=[0, 0], set_max=[1, 1, 1]))
    run("set_gain_dims", lambda: cc.SetConstraint(label="s", expression=q, set_min=[0, 0, 0], set_max=[1, 1, 1], gain=[1.0, 2.0]))
    def set_add_prio():
        a = cc.SetConstraint(label="a", expression=q[0], set_min=0, set_max=1, priority=1); b = cc.SetConstraint(label="b", expression=q[1], set_min=0, set_max=1, priority=2); a + b
    run("set_add_priority", set_add_prio)
    run("velset_dims", lambda: cc.VelocitySetConstraint(label="s", expression=q, set_min=[0, 0], set_max=[1, 1, 1]))
    run("veleq_ok", lambda: cc.VelocityEqualityConstraint(label="v", expression=q, target=[0.0, 0.1, 0.2]))
    run("skill_vel_not_sym", lambda: cc.SkillSpecification("s", t, q, robot_vel_var=2 * dq, constraints=[cc.EqualityConstraint(label="a", expression=q)]))
    run("skill_vel_dims", lambda: cc.SkillSpecification("s", t, q, robot_vel_var=cs.MX.sym("d", 2), constraints=[cc.EqualityConstraint(label="a", expression=q)]))
    run("skill_virt_vel_not_sym", lambda: cc.SkillSpecification("s", t, q, dq, virtual_var=x, virtual_vel_var=2 * dx, constraints=[cc.EqualityConstraint(label="a", expression=q - x)]))
    run("skill_virt_vel_dims", lambda: cc.SkillSpecification("s", t, q, dq, virtual_var=x, virtual_vel_var=cs.MX.sym("d", 2), constraints=[cc.EqualityConstraint(label="a", expression=q - x)]))
    spec = cc.SkillSpecification("s", t, q, dq, constraints=[cc.EqualityConstraint(label="a", expression=q, constraint_type="soft")])
    run("qp_w_matrix", lambda: cc.ReactiveQPController(skill_spec=spec, robot_var_weights=np.eye(3)))
    run("qp_w_dims", lambda: cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0, 2.0]))
    run("qp_w_ok", lambda: cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0, 2.0, 3.0]))
    run("qp_slack_w_dims", lambda: cc.ReactiveQPController(skill_spec=spec, slack_var_weights=[1.0]))
    run("qp_slack_w_matrix", lambda: cc.ReactiveQPController(skill_spec=spec, slack_var_weights=np.eye(3)))
    specv = cc.SkillSpecification("s", t, q, dq, virtual_var=x, virtual_vel_var=dx, constraints=[cc.EqualityConstraint(label="a", expression=q - x, constraint_type="soft")])
    run("qp_virt_w_dims", lambda: cc.ReactiveQPController(skill_spec=specv, virtual_var_weights=[1.0, 2.0]))
    run("qp_virt_w_matrix", lambda: cc.ReactiveQPController(skill_spec=specv, virtual_var_weights=np.eye(2)))
    return out
