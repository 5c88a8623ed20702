"""Two 7-DoF arms in ONE skill: 14 robot variables (casclik/controllers/pseudo_inverse.py:76-88 and reactive_qp.py:191-246
put no bound on the size of robot_var; a dual-arm skill is the reference's own multi-robot use case).  Arm 2 stands
0.8 m beside arm 1; the tools carry a bar between them."""
import numpy as np

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs

BAR = np.array([0.0, 0.8, 0.0])          # tool 2 relative to tool 1, world frame


def frames(fk, q):
    T1 = fk["T_fk"](q[:7])
    T2 = fk["T_fk"](q[7:14])
    p1 = T1[:3, 3]
    p2 = T2[:3, 3] + np.array([0.0, 0.8, 0.0])          # (arm 2's base offset)
    return T1, T2, p1, p2


def two_arm_pinv_skill(fk):
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", 14), cs.MX.sym("y", 3)
    T1, T2, p1, p2 = frames(fk, q)
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    cons = [cc.SetConstraint(label="limits_arm1", expression=q[:7], set_min=lo, set_max=hi, priority=0),
            cc.SetConstraint(label="limits_arm2", expression=q[7:14], set_min=lo, set_max=hi, priority=1),
            cc.EqualityConstraint(label="tool1_to_target", expression=p1 - y, gain=4.0, priority=2),
            cc.EqualityConstraint(label="carry_the_bar", expression=p2 - p1 - BAR, gain=6.0, priority=3),
            cc.EqualityConstraint(label="posture", expression=q - 0.1, gain=0.5, priority=4)]
    return cc.SkillSpecification(label="two_arms", time_var=t, robot_var=q, input_var=y, constraints=cons)


def two_arm_qp_skill(fk):
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", 14), cs.MX.sym("y", 3)
    T1, T2, p1, p2 = frames(fk, q)
    vmax = np.array(list(fk["velocity"]) * 2)
    cons = [cc.EqualityConstraint(label="tool1_to_target", expression=p1 - y, gain=4.0, constraint_type="soft", priority=2),
            cc.EqualityConstraint(label="carry_the_bar", expression=p2 - p1 - BAR, gain=6.0, constraint_type="soft", priority=1),
            cc.VelocitySetConstraint(label="speed_arm1", expression=q[:7], set_min=-vmax[:7], set_max=vmax[:7], priority=0),
            cc.VelocitySetConstraint(label="speed_arm2", expression=q[7:14], set_min=-vmax[7:], set_max=vmax[7:], priority=0)]
    return cc.SkillSpecification(label="two_arms_qp", time_var=t, robot_var=q, input_var=y, constraints=cons)


def two_arm_inputs(fk, B, seed=0, distribution="mixed"):
    Qa, Ya = skills.synthetic_inputs(fk, B, seed=seed, distribution=distribution)
    Qb, _ = skills.synthetic_inputs(fk, B, seed=seed + 101, distribution=distribution)
    return np.hstack([Qa, Qb]), Ya[:, :3].copy()
