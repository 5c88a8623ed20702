"""Skill scripts of the reference-pinned fixtures (tests/golden/ref_pins.npz), written ONCE against
the casclik / casadi API surface and instantiated twice:

  * by tests/golden/make_ref_golden.py with the REFERENCE package (`import casclik as cc` from
    /root/reference) over the stand-in `casadi` module (tests/golden/refshim), forward kinematics
    spelled out with sin / cos / mtimes from the reference's own URDF files;
  * by the tests with the product (`casclik_amd`, `casclik_amd.sym`), forward kinematics from its
    URDF converter.

`env` supplies what differs: cs, cc, T_fk(q) -> 4x4 expression, ori_err(R, quat) -> 3x1 expression,
joint limits / speed limits, and the constants a fixture stores (`consts`).
Reference notebooks these follow: ur5_moe2016_example2.ipynb (1-D and multidimensional limit sets in
front of a pose task), ur5_transformation_matrix...ipynb cell 8 (position task), cart_on_track_1D...
cells 6, 56 (soft pose + speed limits in the QP controller, path following with a virtual variable).
"""
import numpy as np


class Env(object):
    def __init__(self, cs, cc, T_fk, ori_err, lower, upper, vmax, consts=None, T_fk_alt=None):
        self.cs, self.cc, self.T_fk, self.ori_err = cs, cc, T_fk, ori_err
        self.lower, self.upper, self.vmax = np.asarray(lower, float), np.asarray(upper, float), np.asarray(vmax, float)
        self.consts = consts or {}
        self.T_fk_alt = T_fk_alt         # forward kinematics of the OTHER vendored robot (a second chain in one skill)


def _syms(env, ny=0, nx=0):
    cs = env.cs
    n = len(env.lower)
    out = dict(t=cs.MX.sym("t"), q=cs.MX.sym("q", n), dq=cs.MX.sym("dq", n))
    if ny:
        out["y"] = cs.MX.sym("y", ny)
    if nx:
        out["x"] = cs.MX.sym("x", nx)
        out["dx"] = cs.MX.sym("dx", nx)
    return out


def _pose(env, T, p_des, quat_des):
    cs = env.cs
    return cs.vertcat(T[:3, 3] - p_des, env.ori_err(T[:3, :3], quat_des))


def position(env):
    """BASELINE config 1: 3-D tool position, target from input_var"""
    cc, s = env.cc, _syms(env, ny=3)
    T = env.T_fk(s["q"])
    pos = cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - s["y"], gain=10.0,
                                constraint_type="soft", priority=1)
    spec = cc.SkillSpecification(label="position", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[pos])
    return dict(spec=spec, controller="pinv", options={}, ny=3)


def pose(env):
    """BASELINE config 2: 6-D pose task, target from input_var"""
    cc, s = env.cc, _syms(env, ny=7)
    T = env.T_fk(s["q"])
    c = cc.EqualityConstraint(label="tool_pose", expression=_pose(env, T, s["y"][:3], s["y"][3:7]), gain=10.0,
                              constraint_type="soft", priority=1)
    spec = cc.SkillSpecification(label="pose", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[c])
    return dict(spec=spec, controller="pinv", options={}, ny=7)


def _stack(env, multidim, options, time_target=False):
    """BASELINE config 3 with a CONSTANT pose target (the reference cannot build the multidimensional
    tangent-cone function of a skill with an input_var: pseudo_inverse.py:219-221 appends a string to
    the variable list)"""
    cs, cc, s = env.cs, env.cc, _syms(env)
    n = len(env.lower)
    T = env.T_fk(s["q"])
    p_des = np.asarray(env.consts["p_des"], float)
    quat = np.asarray(env.consts["quat_des"], float)
    if time_target:
        p_des = p_des + cs.vertcat(0.05 * cs.sin(s["t"]), 0.02 * s["t"], 0.0)
    mid = 0.5 * (env.lower + env.upper)
    cns = [cc.EqualityConstraint(label="tool_pose", expression=_pose(env, T, p_des, quat), gain=10.0,
                                 constraint_type="soft", priority=10),
           cc.EqualityConstraint(label="joint_centering", expression=s["q"] - mid, gain=1.0,
                                 constraint_type="soft", priority=20)]
    if multidim:
        cns.append(cc.SetConstraint(label="joint_limits", expression=s["q"], set_min=env.lower, set_max=env.upper,
                                    priority=0))
    else:
        # 1-D limit sets on two joints (ur5_moe2016_example2.ipynb cell 6 style), priorities 0 and 1
        for k, j in enumerate((1, 3)):
            cns.append(cc.SetConstraint(label="limit_q%d" % j, expression=s["q"][j], set_min=float(env.lower[j]),
                                        set_max=float(env.upper[j]), priority=k))
    spec = cc.SkillSpecification(label="stack", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 constraints=cns)
    return dict(spec=spec, controller="pinv", options=options, ny=0)


def stack_const(env):
    return _stack(env, True, {"multidim_sets": True})


def stack_const_time(env):
    return _stack(env, True, {"multidim_sets": True}, time_target=True)


def stack_sets1d(env):
    return _stack(env, False, {})


def sets_per_joint(env):
    """one 1-D SetConstraint per joint (ur5_moe2016_example2.ipynb cell 6 does it for the UR5's six) in front of a
    position and a posture task: on the 7-DoF arm 2^7 = 128 modes (pseudo_inverse.py:107-130), scanned in the
    reference's order.  The sets sit at 0.3 of the joint ranges so that the inputs reach many modes."""
    cs, cc, s = env.cs, env.cc, _syms(env)
    n = len(env.lower)
    T = env.T_fk(s["q"])
    p_des = np.asarray(env.consts["p_des"], float)
    cns = [cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - p_des, gain=5.0,
                                 constraint_type="soft", priority=10),
           cc.EqualityConstraint(label="posture", expression=s["q"][n - 2:] - np.array([0.3, -0.2]), gain=1.0,
                                 constraint_type="soft", priority=11)]
    for j in range(n):
        cns.append(cc.SetConstraint(label="limit_q%d" % j, expression=s["q"][j], set_min=float(0.3 * env.lower[j]),
                                    set_max=float(0.3 * env.upper[j]), priority=j))
    spec = cc.SkillSpecification(label="sets_per_joint", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 constraints=cns)
    return dict(spec=spec, controller="pinv", options={}, ny=0)


def stack_noff(env):
    """feedforward off on a time-dependent target"""
    return _stack(env, False, {"feedforward": False}, time_target=True)


def position_standard(env):
    """pinv_method standard (cs.pinv: the undamped normal equations) - a single task: with the doubly
    stacked first equality every later projector of the standard method is singular by construction"""
    out = position(env)
    out["options"] = {"pinv_method": "standard"}
    return out


def conv_last(env):
    """converge_final_set_to_max with the set as the LAST constraint (pseudo_inverse.py:337-379)"""
    cs, cc, s = env.cs, env.cc, _syms(env, ny=3)
    T = env.T_fk(s["q"])
    pos = cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - s["y"], gain=5.0,
                                constraint_type="soft", priority=0)
    lim = cc.SetConstraint(label="limit_q2", expression=s["q"][2], set_min=float(env.lower[2]) * 0.5,
                           set_max=float(env.upper[2]) * 0.5, gain=2.0, priority=5)
    spec = cc.SkillSpecification(label="conv_last", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[pos, lim])
    return dict(spec=spec, controller="pinv", options={"converge_final_set_to_max": True}, ny=3)


def veleq_first(env):
    """a VelocityEqualityConstraint as the first constraint (processed once, :327-335), then an equality"""
    cs, cc, s = env.cs, env.cc, _syms(env, ny=3)
    T = env.T_fk(s["q"])
    vel = cc.VelocityEqualityConstraint(label="tool_z_speed", expression=T[2, 3], target=0.1, priority=0,
                                        constraint_type="soft")
    pos = cc.EqualityConstraint(label="tool_xy", expression=T[:2, 3] - s["y"][:2], gain=4.0,
                                constraint_type="soft", priority=1)
    spec = cc.SkillSpecification(label="veleq_first", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[vel, pos])
    return dict(spec=spec, controller="pinv", options={}, ny=3)


def qp_pose(env):
    """BASELINE config 4: soft pose equality + hard joint-speed VelocitySetConstraint"""
    cs, cc, s = env.cs, env.cc, _syms(env, ny=7)
    T = env.T_fk(s["q"])
    c = cc.EqualityConstraint(label="tool_pose", expression=_pose(env, T, s["y"][:3], s["y"][3:7]), gain=10.0,
                              constraint_type="soft", priority=1)
    speed = cc.VelocitySetConstraint(label="joint_speed_limits", expression=s["q"], set_min=-env.vmax,
                                     set_max=env.vmax, priority=0)
    spec = cc.SkillSpecification(label="qp_pose", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[c, speed])
    return dict(spec=spec, controller="qp", options={}, ny=7)


def qp_limits(env):
    """soft position task + hard joint-limit SetConstraint + hard speed limits (the UR5 notebooks' QP stack)"""
    cs, cc, s = env.cs, env.cc, _syms(env, ny=3)
    T = env.T_fk(s["q"])
    pos = cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - s["y"], gain=3.0,
                                constraint_type="soft", priority=1, slack_weight=2.0)
    lim = cc.SetConstraint(label="joint_limits", expression=s["q"], set_min=env.lower, set_max=env.upper, gain=1.0,
                           priority=0)
    speed = cc.VelocitySetConstraint(label="joint_speed_limits", expression=s["q"], set_min=-env.vmax,
                                     set_max=env.vmax, priority=0)
    spec = cc.SkillSpecification(label="qp_limits", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[pos, lim, speed])
    return dict(spec=spec, controller="qp", options={}, ny=3)


def qp_path(env):
    """path following with a virtual variable (cart_on_track_1D...ipynb cells 56, 75): the tool follows
    p(x) = p0 + x d, the path parameter x is a virtual variable driven to x_goal"""
    cs, cc, s = env.cs, env.cc, _syms(env, nx=1)
    T = env.T_fk(s["q"])
    p0 = np.asarray(env.consts["p_des"], float)
    d = np.array([0.1, -0.05, 0.08])
    follow = cc.EqualityConstraint(label="follow_path", expression=T[:3, 3] - (p0 + d * s["x"]), gain=5.0,
                                   constraint_type="soft", priority=1)
    goal = cc.EqualityConstraint(label="path_goal", expression=s["x"] - 1.0, gain=0.5, constraint_type="soft",
                                 priority=2)
    speed = cc.VelocitySetConstraint(label="joint_speed_limits", expression=s["q"], set_min=-env.vmax,
                                     set_max=env.vmax, priority=0)
    spec = cc.SkillSpecification(label="qp_path", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 virtual_var=s["x"], virtual_vel_var=s["dx"], constraints=[follow, goal, speed])
    return dict(spec=spec, controller="qp", options={}, ny=0, nx=1)


def sym_attrs(env):
    """Gains and set bounds given as EXPRESSIONS of (t, q) - constraints.py:35-39 accepts MX gains, :199-206 MX
    bounds; the controllers multiply / subtract them inside their symbolic expressions
    (pseudo_inverse.py:301-318, :162-185): two 1-D limit sets whose bounds breathe with time and whose gain depends
    on the state, a position task with a time- and state-dependent scalar gain, a centering task with a full
    (n x n) matrix gain whose diagonal depends on the state"""
    cs, cc, s = env.cs, env.cc, _syms(env)
    n = len(env.lower)
    t, q = s["t"], s["q"]
    T = env.T_fk(q)
    p_des = np.asarray(env.consts["p_des"], float) + cs.vertcat(0.05 * cs.sin(t), 0.02 * t, 0.0)
    mid = 0.5 * (env.lower + env.upper)
    kp = 4.0 + cs.sin(0.7 * t) + 0.5 * q[1] * q[1]
    rows = []
    for i in range(n):
        rows.append(cs.horzcat(*[(0.5 + 0.25 * cs.cos(q[i]) if j == i else (0.05 if abs(i - j) == 1 else 0.0))
                                 + 0.0 * q[0] for j in range(n)]))
    Kc = cs.vertcat(*rows)
    cns = [cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - p_des, gain=kp,
                                 constraint_type="soft", priority=10),
           cc.EqualityConstraint(label="joint_centering", expression=q - mid, gain=Kc,
                                 constraint_type="soft", priority=20)]
    for k, j in enumerate((1, 3)):
        shrink = 0.8 + 0.1 * cs.sin(t + k)
        cns.append(cc.SetConstraint(label="limit_q%d" % j, expression=q[j], set_min=float(env.lower[j]) * shrink,
                                    set_max=float(env.upper[j]) * shrink, gain=1.0 + 0.3 * q[0] * q[0], priority=k))
    spec = cc.SkillSpecification(label="sym_attrs", time_var=t, robot_var=q, robot_vel_var=s["dq"], constraints=cns)
    return dict(spec=spec, controller="pinv", options={}, ny=0)


def qp_sym_attrs(env):
    """the same kind of attributes through the QP controller (reactive_qp.py:199-232): soft position task with an
    expression gain, hard joint limits whose bounds breathe with time, speed limits that depend on time, and a
    VelocityEqualityConstraint whose target is an expression"""
    cs, cc, s = env.cs, env.cc, _syms(env, ny=3)
    t, q = s["t"], s["q"]
    n = len(env.lower)
    T = env.T_fk(q)
    kp = 3.0 + cs.sin(0.7 * t) + 0.5 * q[1] * q[1]
    shrink = 0.9 + 0.05 * cs.sin(t)
    pos = cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - s["y"], gain=kp,
                                constraint_type="soft", priority=1, slack_weight=2.0)
    lim = cc.SetConstraint(label="joint_limits", expression=q, set_min=cs.vertcat(*[float(v) * shrink for v in env.lower]),
                           set_max=cs.vertcat(*[float(v) * shrink for v in env.upper]), gain=0.8 + 0.1 * cs.cos(t),
                           priority=0)
    vm = cs.vertcat(*[float(v) * (1.0 + 0.2 * cs.cos(t)) for v in env.vmax])
    speed = cc.VelocitySetConstraint(label="joint_speed_limits", expression=q, set_min=-vm, set_max=vm, priority=0)
    spin = cc.VelocityEqualityConstraint(label="last_joint_spin", expression=q[n - 1], target=0.1 * cs.sin(t) + 0.05 * q[0],
                                         priority=2, constraint_type="soft")
    spec = cc.SkillSpecification(label="qp_sym_attrs", time_var=t, robot_var=q, robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[pos, lim, speed, spin])
    return dict(spec=spec, controller="qp", options={}, ny=3)


def qp_wall(env):
    """a GENERAL inequality row in the QP controller (reactive_qp.py:221-225): a hard SetConstraint on the tool
    height T_fk(q)[2, 3] (the wall sets of ur5_moe2016_example2.ipynb cell 6), a soft set on the tool's x, a soft
    position task and hard joint-speed limits"""
    cs, cc, s = env.cs, env.cc, _syms(env, ny=3)
    T = env.T_fk(s["q"])
    z0 = float(np.asarray(env.consts["p_des"], float)[2])
    x0 = float(np.asarray(env.consts["p_des"], float)[0])
    pos = cc.EqualityConstraint(label="tool_position", expression=T[:3, 3] - s["y"], gain=3.0,
                                constraint_type="soft", priority=5)
    floor = cc.SetConstraint(label="floor", expression=T[2, 3], set_min=z0 - 0.25, set_max=z0 + 0.3, gain=1.0,
                             priority=1)
    fence = cc.SetConstraint(label="fence_x", expression=T[0, 3], set_min=x0 - 0.3, set_max=x0 + 0.35, gain=2.0,
                             priority=2, constraint_type="soft", slack_weight=5.0)
    speed = cc.VelocitySetConstraint(label="joint_speed_limits", expression=s["q"], set_min=-env.vmax,
                                     set_max=env.vmax, priority=0)
    spec = cc.SkillSpecification(label="qp_wall", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 input_var=s["y"], constraints=[pos, floor, fence, speed])
    return dict(spec=spec, controller="qp", options={}, ny=3)


def two_frames(env):
    """several kinematic chains and orientation targets in ONE skill (casclik takes any expression,
    constraints.py:21-24): the iiwa's tool pose, the tool position of a second chain (the UR5) driven by six of the
    same joint variables, a second orientation target on the first tool frame, and a set on a product of an
    orientation-error component with a joint angle plus the squared distance of the two tools"""
    cs, cc, s = env.cs, env.cc, _syms(env)
    q = s["q"]
    T = env.T_fk(q)
    Tb = env.T_fk_alt(cs.vertcat(q[0], q[1], q[2], q[3], q[4], q[5]))
    p_des = np.asarray(env.consts["p_des"], float)
    quat = np.asarray(env.consts["quat_des"], float)
    quat2 = np.array([0.0, 0.6, 0.0, 0.8])
    e1 = env.ori_err(T[:3, :3], quat)
    e2 = env.ori_err(T[:3, :3], quat2)
    dist = Tb[:3, 3] - T[:3, 3]
    cns = [cc.EqualityConstraint(label="tool_pose", expression=cs.vertcat(T[:3, 3] - p_des, e1), gain=5.0,
                                 constraint_type="soft", priority=0),
           cc.EqualityConstraint(label="second_tool", expression=Tb[:3, 3] - cs.vertcat(0.3, -0.2, 0.5), gain=2.0,
                                 constraint_type="soft", priority=1),
           cc.EqualityConstraint(label="second_target", expression=e2, gain=1.5, constraint_type="soft", priority=2),
           cc.SetConstraint(label="mixed", expression=e1[0] * q[0] + cs.mtimes(dist.T, dist), set_min=0.05, set_max=1.2,
                            gain=1.0, priority=3)]
    spec = cc.SkillSpecification(label="two_frames", time_var=s["t"], robot_var=q, robot_vel_var=s["dq"],
                                 constraints=cns)
    return dict(spec=spec, controller="pinv", options={}, ny=0)


def qp_two_virtual(env):
    """a 7-DoF arm with TWO virtual variables (9 states): the tool follows a patch p(x) = p0 + d1 x1 + d2 x2, the patch
    parameters are driven to their goals, joint speeds are limited (cart_on_track_1D...ipynb cells 56, 75 with two
    path parameters)"""
    cs, cc, s = env.cs, env.cc, _syms(env, nx=2)
    T = env.T_fk(s["q"])
    p0 = np.asarray(env.consts["p_des"], float)
    d1, d2 = np.array([0.1, -0.05, 0.08]), np.array([-0.04, 0.09, 0.03])
    follow = cc.EqualityConstraint(label="follow_patch", expression=T[:3, 3] - (p0 + d1 * s["x"][0] + d2 * s["x"][1]),
                                   gain=5.0, constraint_type="soft", priority=1)
    goal = cc.EqualityConstraint(label="patch_goal", expression=s["x"] - np.array([1.0, 0.5]), gain=0.5,
                                 constraint_type="soft", priority=2)
    speed = cc.VelocitySetConstraint(label="joint_speed_limits", expression=s["q"], set_min=-env.vmax,
                                     set_max=env.vmax, priority=0)
    spec = cc.SkillSpecification(label="qp_two_virtual", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"],
                                 virtual_var=s["x"], virtual_vel_var=s["dx"], constraints=[follow, goal, speed])
    return dict(spec=spec, controller="qp", options={}, ny=0, nx=2)


def _heading_rows(env, s):
    """angle-type task errors (constraints.py:21-24 takes any MX; such errors are written with atan2 / acos): the tool's
    heading in the base frame's xy-plane against a target that drifts with time, its elevation through asin, its radius
    held inside a band by fmin / fmax, a saturated (tanh) height error"""
    cs = env.cs
    T = env.T_fk(s["q"])
    p = T[:3, 3]
    heading = cs.atan2(p[1], p[0]) - (0.4 + 0.1 * s["t"])
    radius = cs.sqrt(p[0] * p[0] + p[1] * p[1])
    elevation = cs.asin(p[2] / cs.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + 0.05)) - 0.3
    band = radius - cs.fmin(cs.fmax(radius, 0.35), 0.6)
    tilt = cs.acos(0.9 * T[2, 2]) - 0.8
    height = cs.tanh(3.0 * (p[2] - 0.5))
    bend = cs.atan(2.0 * s["q"][1]) - 0.2
    return cs.vertcat(heading, elevation, band), cs.vertcat(tilt, height, bend)


def heading(env):
    """PseudoInverseController on angle-type errors behind a 1-D joint set"""
    cc, s = env.cc, _syms(env)
    first, second = _heading_rows(env, s)
    cns = [cc.SetConstraint(label="limit_q1", expression=s["q"][1], set_min=float(0.5 * env.lower[1]),
                            set_max=float(0.5 * env.upper[1]), priority=0),
           cc.EqualityConstraint(label="heading", expression=first, gain=2.0, constraint_type="soft", priority=1),
           cc.EqualityConstraint(label="attitude", expression=second, gain=1.0, constraint_type="soft", priority=2)]
    spec = cc.SkillSpecification(label="heading", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"], constraints=cns)
    return dict(spec=spec, controller="pinv", options={}, ny=0)


def qp_heading(env):
    """the same rows under the ReactiveQPController with joint-speed limits"""
    cc, s = env.cc, _syms(env)
    first, second = _heading_rows(env, s)
    cns = [cc.EqualityConstraint(label="heading", expression=first, gain=2.0, constraint_type="soft", priority=1),
           cc.EqualityConstraint(label="attitude", expression=second, gain=1.0, constraint_type="soft", priority=2),
           cc.VelocitySetConstraint(label="joint_speed_limits", expression=s["q"], set_min=-env.vmax, set_max=env.vmax,
                                    priority=0)]
    spec = cc.SkillSpecification(label="qp_heading", time_var=s["t"], robot_var=s["q"], robot_vel_var=s["dq"], constraints=cns)
    return dict(spec=spec, controller="qp", options={}, ny=0)


CASES = {
    "heading": heading, "qp_heading": qp_heading,
    "two_frames": two_frames, "qp_two_virtual": qp_two_virtual, "sets_per_joint": sets_per_joint,
    "stack_boundary": stack_const, "qp_wall": qp_wall,
    "position": position, "pose": pose, "stack_const": stack_const, "stack_const_time": stack_const_time,
    "stack_sets1d": stack_sets1d, "stack_noff": stack_noff, "position_standard": position_standard, "conv_last": conv_last,
    "veleq_first": veleq_first, "qp_pose": qp_pose, "qp_limits": qp_limits, "qp_path": qp_path,
    "sym_attrs": sym_attrs, "qp_sym_attrs": qp_sym_attrs,
}
