"""GPU parity of the PseudoInverseController path: HIP kernel (through the C
ABI) vs the CPU oracles on the same seeded inputs."""
import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from tolerances import PINV_RTOL, PINV_RTOL_TIGHT

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))


def _controller(spec, options=None):
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=None if options is None else dict(options))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    return ctrl


CASES = [
    ("position", skills.position_skill, None, 3),
    ("pose", skills.pose_skill, None, 7),
    ("stack", skills.stack_skill, skills.STACK_OPTIONS, 7),
]


@pytest.mark.parametrize("name,make,options,ny", CASES)
@pytest.mark.parametrize("dist", ["interior", "mixed"])
def test_parity_vs_numpy_oracle(iiwa_fk, name, make, options, ny, dist):
    from oracle import clik_oracle
    spec = make(iiwa_fk)
    ctrl = _controller(spec, options)
    B = 200   # not a multiple of 64: exercises the tail wave
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=3, distribution=dist)
    Y = Y[:, :ny]
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, ref_mode = clik_oracle.pinv_solve_batch(spec, options, 0.0, Q, Y=Y)
    assert np.array_equal(mode, ref_mode)
    err = _rel(dq, ref)
    tol = PINV_RTOL if name == "stack" else 1e-9
    assert err.max() < tol, (name, dist, err.max())


@pytest.mark.parametrize("B", [1, 63, 64, 65, 4096])
def test_parity_vs_c_oracle_sizes(iiwa_fk, B):
    from oracle.c_oracle import CPinvOracle
    spec = skills.stack_skill(iiwa_fk)
    ctrl = _controller(spec, skills.STACK_OPTIONS)
    co = CPinvOracle(spec, skills.STACK_OPTIONS)
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=B, distribution="mixed")
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, _, ref_mode = co.solve_batch(0.0, Q, Y=Y)
    assert np.array_equal(mode, ref_mode)
    assert _rel(dq, ref).max() < PINV_RTOL


def test_single_solve_api(iiwa_fk):
    """Reference call convention: solve(t, q, input_var=y) -> (DM, None, None)."""
    from oracle import clik_oracle
    spec = skills.stack_skill(iiwa_fk)
    ctrl = _controller(spec, skills.STACK_OPTIONS)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 4, seed=11, distribution="mixed")
    for b in range(4):
        res = ctrl.solve(0.0, Q[b], input_var=Y[b])
        assert res[1] is None and res[2] is None
        dq = res[0].toarray()[:, 0]
        ref, ref_mode = clik_oracle.pinv_solve_batch(spec, skills.STACK_OPTIONS, 0.0, Q[b:b + 1], Y=Y[b:b + 1])
        assert ctrl.current_mode == int(ref_mode[0])
        assert _rel(dq[None], ref).max() < PINV_RTOL
