"""Explicit forms of the opaque kinematics atoms of casclik_amd.sym.

casclik takes any CasADi expression as a constraint (casclik/constraints.py:21-24): several kinematic chains in one
skill (two arms, or the elbow and the tool frame of one arm), several orientation targets, an orientation error inside
a larger expression.  The device path keeps ONE chain and ONE orientation target as hand-written kernel code (the
'fk' / 'ori_err' atoms the row table and the generated code read from the kernel's own forward kinematics); everything
beyond that is rewritten here into the explicit expression it stands for - products of the joints' rotation matrices
in sin / cos of the joint variables, the cross products of the orientation error - and then takes the route of any
other expression outside the row table: symbolic differentiation (autodiff.py) and generated device code
(codegen.py), exactly what CasADi's own graph + AD does for the reference (constraints.py:67-73).
"""
from __future__ import annotations

import numpy as np

from . import sym as cs
from .urdf import JOINT_FIXED, JOINT_PRISMATIC, JOINT_REVOLUTE

_FK_CACHE = {}


def _mat(rows):
    return [[r if isinstance(r, cs.Scalar) else cs._c(r) for r in row] for row in rows]


def _matmul(A, B):
    n, k, m = len(A), len(B), len(B[0])
    out = [[cs._ZERO for _ in range(m)] for _ in range(n)]
    for i in range(n):
        for j in range(m):
            acc = cs._ZERO
            for l in range(k):
                acc = cs._s_add(acc, cs._s_mul(A[i][l], B[l][j]))
            out[i][j] = acc
    return out


def fk_entries(chain, qargs):
    """4x4 list of Scalar nodes: the chain's homogeneous transform written out in sin / cos of ``qargs`` (the
    Scalars one 'fk' atom carries), the joints of urdf.Chain in order (T = prod Trans(p) R Rot(axis, q))."""
    key = (id(chain), tuple(id(a) for a in qargs))
    hit = _FK_CACHE.get(key)
    if hit is not None and hit[0] is chain:          # (the entry keeps its chain alive: an id cannot be reused under it)
        return hit[1]
    T = _mat(np.eye(4))
    for j in chain.joints:
        A = np.eye(4)
        A[:3, :3] = j.R
        A[:3, 3] = j.p
        T = _matmul(T, _mat(A))
        if j.type == JOINT_REVOLUTE:
            q = qargs[j.q_index]
            c, s = cs.Scalar("cos", (q,)), cs.Scalar("sin", (q,))
            x, y, z = [float(v) for v in j.axis]
            C = cs._s_sub(cs._ONE, c)

            def e(k0, ks, kc):        # k0 + ks * s + kc * C  with numeric k's (Rodrigues)
                return cs._s_add(cs._s_add(cs._c(k0), cs._s_mul(cs._c(ks), s)), cs._s_mul(cs._c(kc), C))
            R = [[cs._s_add(c, cs._s_mul(cs._c(x * x), C)), e(0.0, -z, x * y), e(0.0, y, x * z)],
                 [e(0.0, z, y * x), cs._s_add(c, cs._s_mul(cs._c(y * y), C)), e(0.0, -x, y * z)],
                 [e(0.0, -y, z * x), e(0.0, x, z * y), cs._s_add(c, cs._s_mul(cs._c(z * z), C))]]
            M = _mat(np.eye(4))
            for a in range(3):
                for b in range(3):
                    M[a][b] = R[a][b]
            T = _matmul(T, M)
        elif j.type == JOINT_PRISMATIC:
            q = qargs[j.q_index]
            M = _mat(np.eye(4))
            for a in range(3):
                M[a][3] = cs._s_mul(cs._c(float(j.axis[a])), q)
            T = _matmul(T, M)
        elif j.type != JOINT_FIXED:
            raise NotImplementedError("joint type %r" % (j.type,))
    if len(_FK_CACHE) > 64:                         # (a few chains per skill: bounded, oldest entries go)
        _FK_CACHE.pop(next(iter(_FK_CACHE)))
    _FK_CACHE[key] = (chain, T)
    return T


def ori_err_entries(rn, qn):
    """The three Scalars of  e_o = 1/2 sum_c r_c x rd_c  (sym.orientation_error) written out: rn = the nine entries
    of R (row-major), qn = the target quaternion (x, y, z, w) as Scalars."""
    x, y, z, w = qn
    m, a, s_ = cs._s_mul, cs._s_add, cs._s_sub
    two = cs._c(2.0)
    one = cs._ONE
    Rd = [[s_(one, m(two, a(m(y, y), m(z, z)))), m(two, s_(m(x, y), m(z, w))), m(two, a(m(x, z), m(y, w)))],
          [m(two, a(m(x, y), m(z, w))), s_(one, m(two, a(m(x, x), m(z, z)))), m(two, s_(m(y, z), m(x, w)))],
          [m(two, s_(m(x, z), m(y, w))), m(two, a(m(y, z), m(x, w))), s_(one, m(two, a(m(x, x), m(y, y))))]]
    R = [[rn[3 * i + j] for j in range(3)] for i in range(3)]
    acc = [cs._ZERO, cs._ZERO, cs._ZERO]
    for c in range(3):
        r = [R[i][c] for i in range(3)]
        d = [Rd[i][c] for i in range(3)]
        cr = [s_(m(r[1], d[2]), m(r[2], d[1])), s_(m(r[2], d[0]), m(r[0], d[2])), s_(m(r[0], d[1]), m(r[1], d[0]))]
        acc = [a(acc[k], cr[k]) for k in range(3)]
    half = cs._c(0.5)
    return [m(half, acc[k]) for k in range(3)]


def rewrite(node, keep_fk, keep_ori, memo):
    """``node`` with every 'fk' atom for which ``keep_fk(node)`` is false and every 'ori_err' atom for which
    ``keep_ori(node)`` is false replaced by its explicit expression (recursively: the R entries of an expanded
    orientation error are rewritten too).  Shared sub-trees stay shared (memo by node identity)."""
    k = id(node)
    if k in memo:
        return memo[k]
    op = node.op
    if op in ("const", "sym"):
        out = node
    elif op == "fk":
        if keep_fk(node):
            out = node
        else:
            chain, i, j = node.aux
            args = tuple(rewrite(a, keep_fk, keep_ori, memo) for a in node.args)
            out = fk_entries(chain, args)[i][j]
    elif op == "ori_err":
        if keep_ori(node):
            out = node
        else:
            args = [rewrite(a, keep_fk, keep_ori, memo) for a in node.args]
            out = ori_err_entries(args[:9], args[9:13])[node.aux]
    else:
        args = tuple(rewrite(a, keep_fk, keep_ori, memo) for a in node.args)
        if all(x is y for x, y in zip(args, node.args)):
            out = node
        else:
            out = cs.Scalar(op, args, value=node.value, name=node.name, index=node.index, family=node.family,
                            aux=node.aux)
    memo[k] = out
    return out


def atoms(node, seen=None, out=None):
    """all 'fk' / 'ori_err' atoms below ``node`` in first-visit order"""
    seen = set() if seen is None else seen
    out = [] if out is None else out
    stack = [node]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if n.op in ("fk", "ori_err"):
            out.append(n)
        stack.extend(reversed(n.args))
    return out
