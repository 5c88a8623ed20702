"""URDF serial-chain reader and the ``converter.from_file`` entry the skill
scripts call.

The reference gets its forward kinematics from the third-party
``urdf2casadi`` package (call sites ``converter.from_file(root=, tip=,
filename=)``: examples/notebooks/ur5_moe2016_example2.ipynb:47,
ur5_transformation_matrix_comparison_of_controllers.ipynb cell 4).  Here the
chain is parsed into plain arrays (per joint: fixed origin transform, axis,
type) that (a) back the symbolic ``T_fk`` atom of casclik_amd.sym and (b) are
copied verbatim into the device skill descriptor, where the HIP kernels run the
FK recursion and build the geometric Jacobian.

URDF convention:  T_i = Trans(xyz) * RPY(rpy) * Rot(axis, q_i).
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET

import numpy as np

from . import sym as _sym

JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_PRISMATIC = 2


def rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr = math.cos(r), math.sin(r)
    cp, sp = math.cos(p), math.sin(p)
    cy, sy = math.cos(y), math.sin(y)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]])


def axis_angle_matrix(axis, angle):
    x, y, z = axis
    c, s = math.cos(angle), math.sin(angle)
    C = 1.0 - c
    return np.array([
        [c + x * x * C, x * y * C - z * s, x * z * C + y * s],
        [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
        [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


class Joint(object):
    __slots__ = ("name", "type", "R", "p", "axis", "lower", "upper",
                 "velocity", "q_index", "rpy")

    def __init__(self, name, jtype, R, p, axis, lower, upper, velocity):
        self.name = name
        self.type = jtype
        self.R = np.asarray(R, dtype=float)
        self.p = np.asarray(p, dtype=float)
        self.axis = np.asarray(axis, dtype=float)
        self.lower = lower
        self.upper = upper
        self.velocity = velocity
        self.q_index = -1
        self.rpy = None       # origin roll-pitch-yaw as written in the URDF (quaternion sign of geom.py)


class Chain(object):
    """Ordered joints from ``root`` to ``tip`` (fixed joints included)."""

    def __init__(self, joints, root="", tip=""):
        self.joints = list(joints)
        self.root = root
        self.tip = tip
        k = 0
        for j in self.joints:
            if j.type != JOINT_FIXED:
                j.q_index = k
                k += 1
        self.n_actuated = k

    @property
    def actuated(self):
        return [j for j in self.joints if j.type != JOINT_FIXED]

    def fk_numeric(self, q):
        q = np.asarray(q, dtype=float).reshape(-1)
        T = np.eye(4)
        for j in self.joints:
            A = np.eye(4)
            A[:3, :3] = j.R
            A[:3, 3] = j.p
            if j.type == JOINT_REVOLUTE:
                M = np.eye(4)
                M[:3, :3] = axis_angle_matrix(j.axis, q[j.q_index])
                A = A.dot(M)
            elif j.type == JOINT_PRISMATIC:
                M = np.eye(4)
                M[:3, 3] = j.axis * q[j.q_index]
                A = A.dot(M)
            T = T.dot(A)
        return T

    def _joint_matrices(self, q):
        """Per joint: (A_fixed, M(q), dM/dq) as 4x4 arrays."""
        out = []
        for j in self.joints:
            A = np.eye(4)
            A[:3, :3] = j.R
            A[:3, 3] = j.p
            M = np.eye(4)
            dM = np.zeros((4, 4))
            if j.type == JOINT_REVOLUTE:
                ang = q[j.q_index]
                M[:3, :3] = axis_angle_matrix(j.axis, ang)
                x, y, z = j.axis
                K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
                dM[:3, :3] = K.dot(M[:3, :3])
            elif j.type == JOINT_PRISMATIC:
                M[:3, 3] = j.axis * q[j.q_index]
                dM[:3, 3] = j.axis
            out.append((A, M, dM))
        return out

    def fk_derivative_numeric(self, q):
        """List over actuated joints k of d T / d q_k (4x4)."""
        q = np.asarray(q, dtype=float).reshape(-1)
        mats = self._joint_matrices(q)
        res = []
        for k in range(self.n_actuated):
            T = np.eye(4)
            for j, (A, M, dM) in zip(self.joints, mats):
                T = T.dot(A).dot(dM if j.q_index == k else M)
            res.append(T)
        return res

    def __call__(self, qvec):
        return _sym.fk_matrix(self, qvec)


def _floats(text, n, default):
    if text is None:
        return list(default)
    vals = [float(v) for v in text.split()]
    if len(vals) != n:
        raise ValueError("expected %d numbers in '%s'" % (n, text))
    return vals


def load_chain(filename, root, tip):
    tree = ET.parse(filename)
    robot = tree.getroot()
    by_child = {}
    for j in robot.findall("joint"):
        by_child[j.find("child").attrib["link"]] = j
    path = []
    link = tip
    while link != root:
        if link not in by_child:
            raise ValueError("no chain from '%s' to '%s' in %s"
                             % (root, tip, filename))
        j = by_child[link]
        path.append(j)
        link = j.find("parent").attrib["link"]
    path.reverse()
    joints = []
    for j in path:
        jtype = j.attrib["type"]
        origin = j.find("origin")
        xyz = _floats(origin.attrib.get("xyz") if origin is not None else None,
                      3, (0.0, 0.0, 0.0))
        rpy = _floats(origin.attrib.get("rpy") if origin is not None else None,
                      3, (0.0, 0.0, 0.0))
        axis_el = j.find("axis")
        axis = _floats(axis_el.attrib.get("xyz") if axis_el is not None else None,
                       3, (1.0, 0.0, 0.0))
        nrm = math.sqrt(sum(a * a for a in axis))
        if nrm > 0:
            axis = [a / nrm for a in axis]
        lim = j.find("limit")
        lower = upper = velocity = None
        if lim is not None:
            lower = float(lim.attrib.get("lower", 0.0))
            upper = float(lim.attrib.get("upper", 0.0))
            velocity = float(lim.attrib["velocity"]) if "velocity" in lim.attrib else None
        if jtype in ("revolute", "continuous"):
            t = JOINT_REVOLUTE
            if jtype == "continuous":
                lower, upper = -math.inf, math.inf
        elif jtype == "prismatic":
            t = JOINT_PRISMATIC
        elif jtype == "fixed":
            t = JOINT_FIXED
        else:
            raise NotImplementedError("joint type '%s'" % jtype)
        joints.append(Joint(j.attrib["name"], t, rpy_matrix(rpy), xyz, axis,
                            lower, upper, velocity))
        joints[-1].rpy = [float(v) for v in rpy]
    return Chain(joints, root, tip)


def chain_from_denavit_hartenberg(joint_angles, link_lengths, link_offsets, link_twists, joint_names=None,
                                  upper_limits=None, lower_limits=None):
    """Serial chain from a classic (distal) Denavit-Hartenberg table,
    ``T_i = Rot_z(theta_i) Trans_z(d_i) Trans_x(a_i) Rot_x(alpha_i)``: ``joint_angles[i]`` is ``"s"`` for an
    actuated (symbolic) revolute joint or a fixed angle (call site: ur5_moe2016_example2.ipynb cell 2, the
    classic DH table of the UR5).  The link transform behind joint i is the origin of joint i+1; the last one
    becomes a fixed joint."""
    n = len(link_lengths)
    if not (len(joint_angles) == len(link_offsets) == len(link_twists) == n):
        raise ValueError("Denavit-Hartenberg lists must have equal length")
    joints = []
    R_prev, p_prev = np.eye(3), np.zeros(3)          # fixed transform accumulated in front of the next joint
    k = 0
    for i in range(n):
        ang = joint_angles[i]
        ca, sa = math.cos(link_twists[i]), math.sin(link_twists[i])
        R_link = np.array([[1.0, 0.0, 0.0], [0.0, ca, -sa], [0.0, sa, ca]])
        p_link = np.array([float(link_lengths[i]), 0.0, float(link_offsets[i])])
        if isinstance(ang, str):
            name = joint_names[k] if joint_names is not None else "joint_%d" % k
            lo = float(lower_limits[k]) if lower_limits is not None else -math.inf
            hi = float(upper_limits[k]) if upper_limits is not None else math.inf
            joints.append(Joint(name, JOINT_REVOLUTE, R_prev, p_prev, [0.0, 0.0, 1.0], lo, hi, None))
            k += 1
            R_prev, p_prev = R_link, p_link
        else:
            Rz = axis_angle_matrix(np.array([0.0, 0.0, 1.0]), float(ang))
            # fold  [R_prev p_prev] Rz [R_link p_link]  into the pending fixed transform
            p_prev = p_prev + R_prev.dot(Rz).dot(p_link)
            R_prev = R_prev.dot(Rz).dot(R_link)
    joints.append(Joint("dh_tip", JOINT_FIXED, R_prev, p_prev, [1.0, 0.0, 0.0], None, None, None))
    return Chain(joints, "dh_base", "dh_tip")


def _fk_dict(chain):
    act = chain.actuated
    q = _sym.MX.sym("q", chain.n_actuated)
    T = chain(q)
    from . import geom
    return {
        "dual_quaternion_fk": _sym.Function("dual_quaternion_fk", [q], [geom.dual_quaternion_fk(chain, q)]),
        "joint_names": [j.name for j in act],
        "upper": [j.upper for j in act],
        "lower": [j.lower for j in act],
        "velocity": [j.velocity for j in act],
        "q": q,
        "T_fk": _sym.Function("T_fk", [q], [T]),
        "chain": chain,
    }


class converter(object):
    """Namespace mirroring ``urdf2casadi.converter``."""

    @staticmethod
    def from_denavit_hartenberg(joint_angles, link_lengths, link_offsets, link_twists, joint_names=None,
                                upper_limits=None, lower_limits=None):
        return _fk_dict(chain_from_denavit_hartenberg(joint_angles, link_lengths, link_offsets, link_twists,
                                                      joint_names, upper_limits, lower_limits))

    @staticmethod
    def from_file(root, tip, filename):
        return _fk_dict(load_chain(filename, root, tip))
