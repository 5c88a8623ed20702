"""GPU: the HIP path (through the controller API -> C ABI) against fixtures produced by the REFERENCE's
own code run over a stand-in casadi (tests/golden/make_ref_golden.py, see tests/test_refpins.py for what
that pins).  Every BASELINE config and every branch of pseudo_inverse.py:274-443 the fixtures cover."""
import numpy as np
import pytest

import casclik_amd as cc
import refpins
from oracle import clik_oracle
from tolerances import QP_RTOL, rtol_from_cond, pinv_close, qp_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", refpins.PINV_NAMES)
def test_pinv_hip_matches_the_reference_run(name):
    built = refpins.product_skill(name)
    Q, Y, X, times = refpins.arrays(name)
    ctrl = cc.PseudoInverseController(skill_spec=built["spec"], options=dict(built["options"]))
    ctrl.setup_problem_functions()
    for ti, t in enumerate(times):
        # (the bound of every instance from the condition numbers the reference's algorithm meets on it - collected by
        # the oracle on the same inputs; what is compared is the REFERENCE's run)
        kappa = np.zeros(len(Q))
        clik_oracle.pinv_solve_batch(built["spec"], built["options"] or None, float(t), Q, Y=Y, cond_out=kappa)
        tol = rtol_from_cond(kappa)
        dq, _, mode = ctrl.solve_batch(float(t), Q, input_var=Y)
        assert np.array_equal(mode, refpins.PINS[name + "_mode"][ti]), (name, ctrl.kernel_name)
        err = refpins.rel_err(dq, refpins.PINS[name + "_dq"][ti])
        assert (err < tol).all(), (name, ctrl.kernel_name, err.max())


@pytest.mark.parametrize("lanes", [1, 4])
def test_stack_kernel_variants_match_the_reference_run(lanes, monkeypatch):
    """both kernel families of the config-3 structure (lane per instance, four lanes per instance)"""
    monkeypatch.setenv("CLIK_LANES", str(lanes))
    # (iiwa_stack_boundary: joints planted exactly on, 5e-13 ... 1e-6 inside and outside of their limits - the
    # 1e-12 thresholds of pseudo_inverse.py:222-252 decide the mode there, and the reference's run is the judge)
    for name in ("iiwa_stack_const", "ur5_stack_const", "iiwa_stack_boundary"):
        built = refpins.product_skill(name)
        Q, Y, X, times = refpins.arrays(name)
        ctrl = cc.PseudoInverseController(skill_spec=built["spec"], options=dict(built["options"]))
        ctrl.setup_problem_functions()
        assert ("/team4" in ctrl.kernel_variant(len(Q))) == (lanes == 4), ctrl.kernel_variant(len(Q))
        dq, _, mode = ctrl.solve_batch(float(times[0]), Q)
        assert np.array_equal(mode, refpins.PINS[name + "_mode"][0])
        assert refpins.rel_err(dq, refpins.PINS[name + "_dq"][0]).max() < 1e-7


@pytest.mark.parametrize("name", refpins.QP_NAMES)
def test_qp_hip_matches_the_reference_run(name):
    built = refpins.product_skill(name)
    Q, Y, X, times = refpins.arrays(name)
    P = refpins.PINS
    t = float(times[0])
    ctrl = cc.ReactiveQPController(skill_spec=built["spec"])
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    if not ctrl.descriptor.extern_code and ctrl.descriptor.n_state <= 8:
        # (H / A / lbA / ubA come from the built-in data kernel; a skill with generated code - here: gains and
        # bounds given as expressions - or with more than 8 states exists only inside the kernel instantiated for
        # it, and its rows are checked through the minimiser below and through the oracle in tests/test_refpins.py)
        H, A, lb, ub = ctrl.qp_data_batch(t, Q, virtual_var=X, input_var=Y)
        assert np.abs(H - P[name + "_H"]).max() < 1e-14
        assert np.abs(A - P[name + "_A"]).max() < 1e-10
        fin = np.abs(P[name + "_lbA"]) < 1e9
        assert np.abs(lb - P[name + "_lbA"])[fin].max() < 1e-9 and np.array_equal(lb[~fin], P[name + "_lbA"][~fin])
        fin = np.abs(P[name + "_ubA"]) < 1e9
        assert np.abs(ub - P[name + "_ubA"])[fin].max() < 1e-9 and np.array_equal(ub[~fin], P[name + "_ubA"][~fin])
    dq, dx, slack, status = ctrl.solve_batch(t, Q, virtual_var=X, input_var=Y)
    assert np.array_equal(status, refpins.ref_status(name)), (name, np.bincount(status))
    ok = status == 0
    assert np.isnan(dq[~ok]).all()
    assert qp_close(dq[ok], P[name + "_dq"][ok])
    assert qp_close(slack[ok], P[name + "_slack"][ok])
    if X is not None:
        assert qp_close(dx[ok], P[name + "_dx"][ok])


@pytest.mark.parametrize("name", refpins.QP_NAMES)
def test_initial_problem_hip_matches_the_reference_run(name):
    """ReactiveQPController.solve_initial_problem against the reference's (reactive_qp.py:300-459): without
    virtual variables the decoupled slack rows, with them the reduced QP solved on the device"""
    built = refpins.product_skill(name)
    Q, Y, X, times = refpins.arrays(name)
    P = refpins.PINS
    ctrl = cc.ReactiveQPController(skill_spec=built["spec"])
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    ctrl.setup_initial_problem_solver()
    for b in range(0, len(Q), 11):
        if refpins.ref_status(name)[b] != 0:
            continue
        virt, slack = ctrl.solve_initial_problem(float(times[0]), Q[b], virtual_var0=None if X is None else X[b],
                                                 input_var0=None if Y is None else Y[b])
        ref = P[name + "_init_slack"][b]
        assert np.abs(np.asarray(slack.toarray()).reshape(-1) - ref).max() < 1e-8 * (1 + np.abs(ref).max())
        if X is not None:
            refv = P[name + "_init_virt"][b]
            assert np.abs(np.asarray(virt.toarray()).reshape(-1) - refv).max() < 1e-8 * (1 + np.abs(refv).max())
        else:
            assert virt is None


@pytest.mark.parametrize("name", refpins.NAMES)
def test_single_instance_solve_returns_what_the_reference_returns(name):
    """`solve()` of one instance through the reference-shaped API (pseudo_inverse.py:512-556, reactive_qp.py:461-528):
    a 3-tuple whose entries are None exactly where the reference's are (no virtual variable / no slack) and (n, 1)
    matrices elsewhere (fixture `<name>_ret`, recorded from the reference's own return values), with the first
    instance's numbers equal to the batch path's."""
    built = refpins.product_skill(name)
    spec = built["spec"]
    Q, Y, X, times = refpins.arrays(name)
    if built["controller"] == "pinv":
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(built["options"]) if built["options"] else None)
    else:
        ctrl = cc.ReactiveQPController(skill_spec=spec, options=dict(built["options"]) if built["options"] else None)
    try:
        ctrl.setup_problem_functions()
    except NotImplementedError:
        pytest.skip("skill outside the kernels of this build (covered by its own test)")
    ctrl.setup_solver()
    has_y = spec.input_var is not None and spec._has_input
    kw = {}
    if has_y:
        kw["input_var"] = Y[0]
    if X is not None:
        kw["virtual_var"] = X[0]
    try:
        res = ctrl.solve(float(times[0]), Q[0], **kw)
    except RuntimeError:
        assert refpins.ref_status(name)[0] == 2            # (the reference raises on its infeasible instances too)
        return
    want = refpins.PINS[name + "_ret"]
    assert isinstance(res, tuple) and len(res) == len(want)
    got = [-1 if r is None else r.toarray().shape[0] * 100 + r.toarray().shape[1] for r in res]
    assert got == want.tolist(), (got, want.tolist())
