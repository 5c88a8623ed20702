"""Skill scripts for the BASELINE.json configurations.

Written the way a casclik user writes a skill (cf. reference notebooks
examples/notebooks/ur5_moe2016_example2.ipynb cells 4-8,
ur5_transformation_matrix_comparison_of_controllers.ipynb cell 8): symbols,
``T_fk`` from the URDF converter, constraint objects, a SkillSpecification.
Robot: KUKA LBR iiwa 14 R820, chain base_link -> tool0 (7 DoF).
"""
from __future__ import annotations

import os

import numpy as np

from . import sym as cs
from .constraints import (EqualityConstraint, SetConstraint,
                          VelocitySetConstraint)
from .skill_specification import SkillSpecification
from .urdf import converter

ROBOT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")
IIWA_URDF = os.path.join(ROBOT_DIR, "lbr_iiwa_14_r820.urdf")
UR5_URDF = os.path.join(ROBOT_DIR, "ur5.urdf")


def iiwa():
    return converter.from_file(root="base_link", tip="tool0", filename=IIWA_URDF)


def ur5():
    return converter.from_file(root="base_link", tip="tool0", filename=UR5_URDF)


def position_skill(fk=None):
    """Config 1: one 3-D end-effector EqualityConstraint, target from
    input_var y[0:3]."""
    fk = fk or iiwa()
    n = len(fk["joint_names"])
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    dq = cs.MX.sym("dq", n)
    y = cs.MX.sym("y", 3)
    T = fk["T_fk"](q)
    pos = EqualityConstraint(label="tool_position", expression=T[:3, 3] - y,
                             gain=10.0, constraint_type="soft", priority=1)
    return SkillSpecification(label="position", time_var=t, robot_var=q,
                              robot_vel_var=dq, input_var=y, constraints=[pos])


def _pose_expression(T, y):
    return cs.vertcat(T[:3, 3] - y[:3], cs.orientation_error(T[:3, :3], y[3:7]))


def pose_skill(fk=None, gain=10.0):
    """Config 2: single 6-D pose EqualityConstraint
    e = [p_fk - p_des ; 1/2 sum_i r_i x r_i,des], per-instance target
    y = [p_des(3), quat_des(4, xyzw)], gain 10."""
    fk = fk or iiwa()
    n = len(fk["joint_names"])
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    dq = cs.MX.sym("dq", n)
    y = cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    pose = EqualityConstraint(label="tool_pose", expression=_pose_expression(T, y),
                              gain=gain, constraint_type="soft", priority=1)
    return SkillSpecification(label="pose", time_var=t, robot_var=q,
                              robot_vel_var=dq, input_var=y, constraints=[pose])


def stack_skill(fk=None):
    """Config 3: priority stack  [multidim joint-limit SetConstraint ;
    6-D pose EqualityConstraint ; joint-centering EqualityConstraint]."""
    fk = fk or iiwa()
    n = len(fk["joint_names"])
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    dq = cs.MX.sym("dq", n)
    y = cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    q_min = np.array(fk["lower"])
    q_max = np.array(fk["upper"])
    q_mid = 0.5 * (q_min + q_max)
    limits = SetConstraint(label="joint_limits", expression=q, set_min=q_min,
                           set_max=q_max, priority=0)
    pose = EqualityConstraint(label="tool_pose", expression=_pose_expression(T, y),
                              gain=10.0, constraint_type="soft", priority=1)
    center = EqualityConstraint(label="joint_centering", expression=q - q_mid,
                                gain=1.0, constraint_type="soft", priority=2)
    return SkillSpecification(label="stack", time_var=t, robot_var=q,
                              robot_vel_var=dq, input_var=y,
                              constraints=[pose, center, limits])


def qp_skill(fk=None):
    """Config 4: soft pose equality + joint-velocity VelocitySetConstraint
    (13 variables x 13 rows on the iiwa)."""
    fk = fk or iiwa()
    n = len(fk["joint_names"])
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    dq = cs.MX.sym("dq", n)
    y = cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    v_max = np.array(fk["velocity"])
    pose = EqualityConstraint(label="tool_pose", expression=_pose_expression(T, y),
                              gain=10.0, constraint_type="soft", priority=1)
    speed = VelocitySetConstraint(label="joint_speed_limits", expression=q,
                                  set_min=-v_max, set_max=v_max, priority=0)
    return SkillSpecification(label="qp_pose", time_var=t, robot_var=q,
                              robot_vel_var=dq, input_var=y,
                              constraints=[pose, speed])


STACK_OPTIONS = {"multidim_sets": True}


def quat_from_matrix(R):
    """Unit quaternion (x, y, z, w) of a rotation matrix (Shepperd)."""
    R = np.asarray(R, dtype=float)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    return np.array([x, y, z, w])


def synthetic_inputs(fk, B, seed=0, distribution="interior"):
    """Seeded synthetic batch: joint states Q [B,n] and pose
    targets Y [B,7] = [p_des, quat_des(xyzw)] from FK(q_des).

    interior: q ~ U(0.9 lo, 0.9 hi);  mixed: q ~ U(lo - 0.05 r, hi + 0.05 r)
    (exercises the joint-limit set);  q_des ~ U(0.8 lo, 0.8 hi)."""
    rng = np.random.default_rng(seed)
    lo = np.array(fk["lower"], dtype=float)
    hi = np.array(fk["upper"], dtype=float)
    n = lo.size
    if distribution == "interior":
        Q = rng.uniform(0.9 * lo, 0.9 * hi, size=(B, n))
    elif distribution == "mixed":
        r = hi - lo
        Q = rng.uniform(lo - 0.05 * r, hi + 0.05 * r, size=(B, n))
    else:
        raise ValueError("distribution must be 'interior' or 'mixed'")
    Qd = rng.uniform(0.8 * lo, 0.8 * hi, size=(B, n))
    chain = fk["chain"]
    Y = np.zeros((B, 7))
    for b in range(B):
        T = chain.fk_numeric(Qd[b])
        Y[b, :3] = T[:3, 3]
        Y[b, 3:] = quat_from_matrix(T[:3, :3])
    return Q, Y
