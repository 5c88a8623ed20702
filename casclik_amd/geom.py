"""Quaternion / dual-quaternion helpers with the call surface of
``urdf2casadi.casadi_geom`` and ``urdf2casadi.numpy_geom`` as the reference
notebooks use them (ur5_dual_quaternion_comparison_of_controllers.ipynb cells
4-7, 33; ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 3, 10, 16-18),
and the dual-quaternion forward kinematics behind ``fk_dict["dual_quaternion_fk"]``.

urdf2casadi is a third-party dependency that is not part of the reference tree;
this restates the standard algebra in the layout its printed outputs pin:
a dual quaternion is 8 numbers ``[x, y, z, w | x', y', z', w']`` (real part
first, scalar last in each half), a rigid transform (R, t) maps to
``[r ; 1/2 t (x) r]``: ``dual_quaternion_revolute([.2,.2,.75],[0,0,0],[1,0,0],0)
= [0,0,0,1, .1,.1,.375,0]`` (cell 33) and ``Q_fk(UR5_home)`` of cell 7 are the
known answers in tests/test_oracle.py.

The same code serves numbers (numpy in, numpy out) and expressions
(casclik_amd.sym.MX in, MX out): it only uses + - * on the entries.
"""
from __future__ import annotations

import math

import numpy as np

from . import sym as _sym


def _entries(v, n):
    if isinstance(v, _sym.MX):
        a = _sym._as_array(v).T.reshape(-1)
        if a.size != n:
            raise ValueError("expected %d entries, got %d" % (n, a.size))
        return [_sym._wrap(np.array([[s]], dtype=object)) for s in a], True
    if isinstance(v, _sym.DM):
        v = v.toarray()
    a = np.asarray(v, dtype=float).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d entries, got %d" % (n, a.size))
    return list(a), False


def _pack(items, symbolic):
    if symbolic or any(isinstance(i, _sym.MX) for i in items):
        return _sym.vertcat(*items)
    return np.array(items, dtype=float)


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz]


def quaternion_product(quat1, quat2):
    a, s1 = _entries(quat1, 4)
    b, s2 = _entries(quat2, 4)
    return _pack(_qmul(a, b), s1 or s2)


def quaternion_conj(quat):
    a, s = _entries(quat, 4)
    return _pack([-a[0], -a[1], -a[2], a[3]], s)


def dual_quaternion_product(Q1, Q2):
    a, s1 = _entries(Q1, 8)
    b, s2 = _entries(Q2, 8)
    real = _qmul(a[:4], b[:4])
    d1 = _qmul(a[:4], b[4:])
    d2 = _qmul(a[4:], b[:4])
    return _pack(real + [d1[i] + d2[i] for i in range(4)], s1 or s2)


def dual_quaternion_conj(Q):
    a, s = _entries(Q, 8)
    return _pack([-a[0], -a[1], -a[2], a[3], -a[4], -a[5], -a[6], a[7]], s)


def dual_quaternion_norm2(Q):
    """The dual number Q (x) conj(Q) = |r|^2 + eps 2 r.d as [real, dual] (unpinned: the notebooks only
    wrap it into a Function, ur5_dual_quaternion_vs_transformation_matrix.ipynb cell 3)."""
    a, s = _entries(Q, 8)
    rr = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]
    rd = a[0] * a[4] + a[1] * a[5] + a[2] * a[6] + a[3] * a[7]
    return _pack([rr, 2.0 * rd], s)


def dual_quaternion_inv(Q):
    """Inverse of a dual quaternion with non-zero real part: r^-1 - eps r^-1 d r^-1."""
    a, s = _entries(Q, 8)
    rr = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]
    ri = [-a[0] / rr, -a[1] / rr, -a[2] / rr, a[3] / rr]
    m = _qmul(_qmul(ri, a[4:]), ri)
    return _pack(ri + [-m[0], -m[1], -m[2], -m[3]], s)


def dual_quaternion_to_pos(Q):
    """Translation 2 d (x) conj(r) of a unit dual quaternion."""
    a, s = _entries(Q, 8)
    tq = _qmul(a[4:], [-a[0], -a[1], -a[2], a[3]])
    return _pack([2.0 * tq[0], 2.0 * tq[1], 2.0 * tq[2]], s)


def quaternion_rpy(roll, pitch, yaw):
    """Quaternion of the URDF fixed-axis rotation Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr = math.cos(0.5 * roll), math.sin(0.5 * roll)
    cp, sp = math.cos(0.5 * pitch), math.sin(0.5 * pitch)
    cy, sy = math.cos(0.5 * yaw), math.sin(0.5 * yaw)
    return np.array([sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy])


def _trig(v):
    """(sin, cos) of a number or an expression."""
    if isinstance(v, _sym.MX):
        return _sym.sin(v), _sym.cos(v)
    return math.sin(float(v)), math.cos(float(v))


def dual_quaternion_rpy(rpy):
    """Pure rotation Rz(yaw) Ry(pitch) Rx(roll); the angles may be expressions."""
    a, sym = _entries(rpy, 3)
    (sr, cr), (sp, cp), (sy, cy) = _trig(0.5 * a[0]), _trig(0.5 * a[1]), _trig(0.5 * a[2])
    r = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
         cr * cp * cy + sr * sp * sy]
    return _pack(r + [0.0, 0.0, 0.0, 0.0], sym)


def dual_quaternion_translation(xyz):
    a, sym = _entries(xyz, 3)
    return _pack([0.0, 0.0, 0.0, 1.0, 0.5 * a[0], 0.5 * a[1], 0.5 * a[2], 0.0], sym)


def dual_quaternion_axis_translation(axis, ang):
    a, s1 = _entries(axis, 3)
    g, s2 = _entries(ang, 1)
    return _pack([0.0, 0.0, 0.0, 1.0, 0.5 * a[0] * g[0], 0.5 * a[1] * g[0], 0.5 * a[2] * g[0], 0.0], s1 or s2)


def dual_quaternion_axis_rotation(axis, ang):
    a, s1 = _entries(axis, 3)
    g, s2 = _entries(ang, 1)
    sn, cn = _trig(0.5 * g[0])
    return _pack([a[0] * sn, a[1] * sn, a[2] * sn, cn, 0.0, 0.0, 0.0, 0.0], s1 or s2)


def T_rpy(displacement, roll, pitch, yaw):
    """Homogeneous transform with translation ``displacement`` and rotation RPY."""
    T = np.eye(4)
    T[:3, :3] = rotation_rpy(roll, pitch, yaw)
    T[:3, 3] = np.asarray(displacement, dtype=float).reshape(-1)
    return T


def rotation_rpy(roll, pitch, yaw):
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _axis_quaternion(axis, qi):
    """[axis sin(qi/2), cos(qi/2)]; qi may be an expression."""
    ax = np.asarray(axis, dtype=float).reshape(-1)
    nrm = np.linalg.norm(ax)
    ax = ax / nrm if nrm > 0 else ax
    if isinstance(qi, _sym.MX):
        s, c = _sym.sin(0.5 * qi), _sym.cos(0.5 * qi)
    else:
        s, c = math.sin(0.5 * float(qi)), math.cos(0.5 * float(qi))
    return [float(ax[0]) * s, float(ax[1]) * s, float(ax[2]) * s, c]


def _rigid(r, t):
    """Dual quaternion [r ; 1/2 t (x) r] of rotation quaternion r and translation t."""
    tq = [t[0], t[1], t[2], 0.0]
    d = _qmul(tq, r)
    return list(r) + [0.5 * d[i] for i in range(4)]


def dual_quaternion_revolute(xyz, rpy, axis, qi):
    """Trans(xyz) RPY(rpy) Rot(axis, qi) - one URDF revolute joint."""
    r = _qmul(list(quaternion_rpy(*[float(v) for v in rpy])), _axis_quaternion(axis, qi))
    sym = isinstance(qi, _sym.MX)
    return _pack(_rigid(r, [float(v) for v in xyz]), sym)


def dual_quaternion_prismatic(xyz, rpy, axis, qi):
    """Trans(xyz) RPY(rpy) Trans(axis qi) - one URDF prismatic joint."""
    r0 = list(quaternion_rpy(*[float(v) for v in rpy]))
    R0 = rotation_rpy(*[float(v) for v in rpy])
    ax = R0.dot(np.asarray(axis, dtype=float).reshape(-1))
    t = [float(xyz[i]) + float(ax[i]) * qi for i in range(3)]
    return _pack(_rigid(r0, t), isinstance(qi, _sym.MX))


def dual_quaternion_to_transformation_matrix(Q):
    a, sym = _entries(Q, 8)
    x, y, z, w = a[:4]
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    tq = _qmul(a[4:], [-x, -y, -z, w])
    t = [2.0 * tq[0], 2.0 * tq[1], 2.0 * tq[2]]
    if sym:
        rows = [_sym.horzcat(R[i][0], R[i][1], R[i][2], t[i]) for i in range(3)]
        rows.append(_sym.horzcat(0.0, 0.0, 0.0, 1.0))
        return _sym.vertcat(*rows)
    T = np.eye(4)
    T[:3, :3] = np.array(R, dtype=float)
    T[:3, 3] = t
    return T


def dual_quaternion_fk(chain, qvec):
    """Dual quaternion of the chain's tool frame as an expression of the joint variables: the
    product of the per-joint dual quaternions (fixed joints fold into constants)."""
    from .urdf import JOINT_FIXED, JOINT_PRISMATIC
    qa = _sym._as_array(qvec).reshape(-1)
    if len(qa) != chain.n_actuated:
        raise ValueError("dual_quaternion_fk expects %d joint values, got %d" % (chain.n_actuated, len(qa)))
    Q = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0])
    for j in chain.joints:
        rpy = j.rpy if getattr(j, "rpy", None) is not None else _rpy_of(j.R)
        if j.type == JOINT_FIXED:
            Qj = dual_quaternion_revolute(j.p, rpy, [1.0, 0.0, 0.0], 0.0)
        else:
            qi = _sym._wrap(np.array([[qa[j.q_index]]], dtype=object))
            if qi.is_constant():
                qi = float(_sym.evaluate(qi, {})[0, 0])
            Qj = (dual_quaternion_prismatic if j.type == JOINT_PRISMATIC else dual_quaternion_revolute)(
                j.p, rpy, j.axis, qi)
        Q = dual_quaternion_product(Q, Qj)
    return Q


def _rpy_of(R):
    """Fixed-axis roll-pitch-yaw of a rotation matrix (inverse of rotation_rpy away from pitch = +-90 deg,
    any valid triple there)."""
    R = np.asarray(R, dtype=float).reshape(3, 3)
    sp = -R[2, 0]
    if abs(sp) < 1.0 - 1e-12:
        return [math.atan2(R[2, 1], R[2, 2]), math.asin(sp), math.atan2(R[1, 0], R[0, 0])]
    # gimbal lock: yaw := 0
    pitch = math.copysign(0.5 * math.pi, sp)
    return [math.atan2(-R[1, 2], R[1, 1]), pitch, 0.0]


class _Namespace(object):
    """Module-like holder (``from casclik_amd import casadi_geom, numpy_geom``)."""

    def __init__(self, name, **fns):
        self.__name__ = name
        for k, v in fns.items():
            setattr(self, k, v)


_COMMON = dict(quaternion_product=quaternion_product, quaternion_conj=quaternion_conj,
               dual_quaternion_product=dual_quaternion_product, dual_quaternion_conj=dual_quaternion_conj,
               dual_quaternion_norm2=dual_quaternion_norm2, dual_quaternion_inv=dual_quaternion_inv,
               dual_quaternion_to_pos=dual_quaternion_to_pos,
               dual_quaternion_revolute=dual_quaternion_revolute,
               dual_quaternion_prismatic=dual_quaternion_prismatic,
               dual_quaternion_to_transformation_matrix=dual_quaternion_to_transformation_matrix,
               quaternion_rpy=quaternion_rpy, rotation_rpy=rotation_rpy, T_rpy=T_rpy,
               dual_quaternion_rpy=dual_quaternion_rpy, dual_quaternion_translation=dual_quaternion_translation,
               dual_quaternion_axis_translation=dual_quaternion_axis_translation,
               dual_quaternion_axis_rotation=dual_quaternion_axis_rotation)
casadi_geom = _Namespace("casadi_geom", **_COMMON)
numpy_geom = _Namespace("numpy_geom", **_COMMON)
