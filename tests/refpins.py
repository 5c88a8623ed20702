"""Shared helpers of the reference-pinned tests (tests/golden/ref_pins.npz, made by
tests/golden/make_ref_golden.py from the reference's own Python over a stand-in casadi)."""
import os

import numpy as np

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs

HERE = os.path.dirname(os.path.abspath(__file__))
import sys
sys.path.insert(0, os.path.join(HERE, "golden"))
import pin_skills      # noqa: E402

PINS = np.load(os.path.join(HERE, "golden", "ref_pins.npz"))
NAMES = sorted({k[:-2] for k in PINS.files if k.endswith("_Q")})
PINV_NAMES = [n for n in NAMES if n + "_mode" in PINS.files]
QP_NAMES = [n for n in NAMES if n + "_H" in PINS.files]
_FK = {}


def robot_fk(name):
    robot = name.split("_", 1)[0]
    if robot not in _FK:
        _FK[robot] = skills.iiwa() if robot == "iiwa" else skills.ur5()
    return _FK[robot]


def product_skill(name):
    """the fixture's skill built with the product's front-end (same script as the reference side)"""
    fk = robot_fk(name)
    other = robot_fk("ur5_x" if name.startswith("iiwa") else "iiwa_x")
    env = pin_skills.Env(cs, cc, fk["T_fk"], cs.orientation_error, fk["lower"], fk["upper"], fk["velocity"],
                         {"p_des": PINS[name + "_p_des"], "quat_des": PINS[name + "_quat_des"]},
                         T_fk_alt=other["T_fk"])
    return pin_skills.CASES[name.split("_", 1)[1]](env)


def arrays(name):
    get = lambda k: PINS[name + "_" + k] if (name + "_" + k) in PINS.files else None   # noqa: E731
    return get("Q"), get("Y"), get("X"), PINS[name + "_t"]


def ref_status(name):
    """status per instance of a QP fixture: 2 where the reference's solve() raised (infeasible rows, confirmed by an
    LP in the generator), 0 elsewhere"""
    key = name + "_status"
    return PINS[key] if key in PINS.files else np.zeros(len(PINS[name + "_Q"]), dtype=np.int32)


def sigma_min_geometric(fk, Q):
    """smallest singular value of the chain's geometric Jacobian [Jv; Jw] per instance"""
    chain = fk["chain"]
    out = np.zeros(len(Q))
    for b, q in enumerate(Q):
        T = chain.fk_numeric(q)
        cols = []
        for dT in chain.fk_derivative_numeric(q):
            W = dT[:3, :3].dot(T[:3, :3].T)          # skew(omega_k)
            cols.append(np.concatenate([dT[:3, 3], [W[2, 1], W[0, 2], W[1, 0]]]))
        out[b] = np.linalg.svd(np.array(cols).T, compute_uv=False)[-1]
    return out


def rel_err(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))
