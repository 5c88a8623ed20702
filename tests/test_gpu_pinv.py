"""GPU parity of the PseudoInverseController path: HIP kernel (through the C
ABI) vs the CPU oracles on the same seeded inputs."""
import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs
from tolerances import PINV_RTOL, PINV_RTOL_TIGHT, pinv_close, qp_close

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
    assert pinv_close(dq, ref), (name, dist, _rel(dq, ref).max())


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
    assert pinv_close(dq, ref)


def test_golden_vectors(iiwa_fk, ur5_fk):
    """HIP vs the committed golden fixtures (tests/golden/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clik_golden.npz"))
    for name, fk, make, opts, tol in [
            ("iiwa_position", iiwa_fk, skills.position_skill, None, 1e-9),
            ("iiwa_pose", iiwa_fk, skills.pose_skill, None, 1e-9),
            ("iiwa_stack", iiwa_fk, skills.stack_skill, skills.STACK_OPTIONS, PINV_RTOL),
            ("ur5_stack", ur5_fk, skills.stack_skill, skills.STACK_OPTIONS, PINV_RTOL)]:
        ctrl = _controller(make(fk), opts)
        dq, _, mode = ctrl.solve_batch(0.0, g[name + "_Q"], input_var=g[name + "_Y"])
        assert np.array_equal(mode, g[name + "_mode"]), name
        assert _rel(dq, g[name + "_dq"]).max() < tol, name


def test_static_and_dynamic_kernels_agree(iiwa_fk, monkeypatch):
    """The AOT shape-specialised kernel and the dynamic-shape kernel are two
    instantiations of the same algebra: same modes, same velocities."""
    spec = skills.stack_skill(iiwa_fk)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 300, seed=21, distribution="mixed")
    fast = _controller(spec, skills.STACK_OPTIONS)
    assert fast.kernel_name == "kStackIiwa"
    monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    slow = _controller(spec, skills.STACK_OPTIONS)
    assert slow.kernel_name == "dynamic"
    a, _, ma = fast.solve_batch(0.0, Q, input_var=Y)
    b, _, mb = slow.solve_batch(0.0, Q, input_var=Y)
    assert np.array_equal(ma, mb)
    assert pinv_close(a, b)


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_skills_without_aot_shape(ur5_fk, kernel, monkeypatch):
    """Skills with no AOT shape: 1-D sets with a 32-mode scan + scalar distance
    task (ur5_transformation_matrix... cell 27) and a time-trajectory tracking
    task with feed-forward (ur5_moe2016_example2 cell 7); served by run-time
    instantiated kernels (32 mode bodies) or by the dynamic-shape kernels."""
    from oracle import clik_oracle
    from casclik_amd import sym as cs
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    cons = [cc.EqualityConstraint("dist", cs.norm_2(np.array([0.5, 0.5, 0.5]) - p), gain=50.0,
                                  constraint_type="soft", priority=6)]
    for i in range(5):
        cons.append(cc.SetConstraint("limit_q_%d" % i, q[i], set_min=0.3 * lo[i], set_max=0.3 * hi[i], priority=i))
    spec = cc.SkillSpecification("point", t, q, constraints=cons)
    ctrl = _controller(spec)
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic") and ctrl.n_modes == 32
    rng = np.random.default_rng(4)
    Q = rng.uniform(0.35 * lo, 0.35 * hi, size=(130, 6))
    dq, _, mode = ctrl.solve_batch(0.0, Q)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q)
    assert np.array_equal(mode, rmode) and len(np.unique(mode)) > 3
    assert pinv_close(dq, ref)

    path = cs.vertcat(0.5 * cs.sin(0.1 * t) * cs.sin(0.1 * t) + 0.2, 0.5 * cs.cos(0.1 * t) + 0.25 * cs.sin(0.1 * t),
                      0.5 * cs.sin(0.1 * t) * cs.cos(0.1 * t) + 0.1)
    track = cc.SkillSpecification("track", t, q, constraints=[
        cc.EqualityConstraint("move_point", p - path, gain=0.15, constraint_type="soft")])
    tc = _controller(track)
    for tval in (0.0, 7.3):
        dq, _, mode = tc.solve_batch(tval, Q)
        ref, _ = clik_oracle.pinv_solve_batch(track, None, tval, Q)
        assert _rel(dq, ref).max() < 1e-9


def test_rollout_matches_host_loop(iiwa_fk):
    """n ticks of solve -> clamp -> Euler in one launch == the host loop of
    ur5_moe2016_example2.ipynb:537-545 driven tick by tick."""
    spec = skills.stack_skill(iiwa_fk)
    ctrl = _controller(spec, skills.STACK_OPTIONS)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 100, seed=13, distribution="mixed")
    dt, vmax, n_ticks = 0.008, np.pi / 5, 12
    q = Q.copy()
    for _ in range(n_ticks):
        dq, _, mode = ctrl.solve_batch(0.0, q, input_var=Y)
        dq = np.clip(dq, -vmax, vmax)
        q = q + dq * dt
    q_dev, dq_dev, mode_dev = ctrl.rollout_batch(np.zeros(n_ticks), Q, input_var=Y, dt=dt, max_speed=vmax)
    assert np.array_equal(mode_dev, mode)
    assert np.abs(q_dev - q).max() < 1e-9 and np.abs(dq_dev - dq).max() < 1e-7


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
        assert pinv_close(dq[None], ref)


def test_role_split_kernel_matches_oracle(iiwa_fk, monkeypatch):
    """Opt-in four-wave kernel (main + helper wave per mode, CLIK_ROLE_SPLIT=1): the helper
    builds and factors the Gram-form stack, the main wave projects through it."""
    from oracle import clik_oracle
    monkeypatch.setenv("CLIK_ROLE_SPLIT", "1")
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    Q, Y = skills.synthetic_inputs(iiwa_fk, 200, seed=31, distribution="mixed")
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, dict(skills.STACK_OPTIONS), 0.0, Q, Y=Y)
    assert np.array_equal(mode, rmode) and len(np.unique(mode)) >= 2
    assert pinv_close(dq, ref)


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_huge_joint_angles_take_the_accurate_sincos_path(iiwa_fk, kernel, monkeypatch):
    """Joint angles beyond the fast range reduction (|q| > 1e5 rad, a few lanes per wave)
    go through the library-accurate sin/cos in one cold block; results still match the
    oracle (which uses libm for every angle)."""
    from oracle import clik_oracle
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    spec = skills.pose_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    Q, Y = skills.synthetic_inputs(iiwa_fk, 130, seed=5)
    rng = np.random.default_rng(6)
    rows = rng.choice(130, size=12, replace=False)
    Q[rows, rng.integers(0, 7, size=12)] += rng.choice([-1.0, 1.0], size=12) * 2 * np.pi * rng.integers(2e4, 3e6, size=12)
    assert (np.abs(Q) > 1e5).any()
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q, Y=Y)
    assert np.array_equal(mode, rmode)
    assert pinv_close(dq, ref)


@pytest.mark.parametrize("skill", ["pose", "stack"])
def test_near_singular_configurations(iiwa_fk, skill):
    """Stretched-arm configurations (q ~ 0: the iiwa's elbow/wrist singularity) where the damped
    inverse is doing real work (smallest singular value^2 comparable to lam = 1e-7 or a few
    orders above): the fused double processing of the first equality and the push-through
    projection must still agree with the literal algorithm within PINV_RTOL."""
    from oracle import clik_oracle
    if skill == "pose":
        spec, opts = skills.pose_skill(iiwa_fk), None
    else:
        spec, opts = skills.stack_skill(iiwa_fk), dict(skills.STACK_OPTIONS)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
    ctrl.setup_problem_functions()
    rng = np.random.default_rng(41)
    _, Y = skills.synthetic_inputs(iiwa_fk, 192, seed=42)
    scales = np.repeat([1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 0.0], 32)[:, None]
    Q = scales * rng.normal(size=(192, 7))
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, 0.0, Q, Y=Y)
    assert np.isfinite(dq).all()
    assert np.array_equal(mode, rmode)
    assert pinv_close(dq, ref), _rel(dq, ref).max()


@pytest.mark.parametrize("B", [1, 63, 64, 65, 129])
def test_ragged_batch_sizes(iiwa_fk, B):
    """Tail blocks (rows_valid < 64) of the two-wave kernel, the one-wave kernel and the
    shape-specialised QP kernel: results identical to the same rows of a larger batch."""
    from oracle import clik_oracle
    Qall, Yall = skills.synthetic_inputs(iiwa_fk, 192, seed=77, distribution="mixed")
    Q, Y = Qall[:B], Yall[:B]
    for spec, opts in ((skills.stack_skill(iiwa_fk), dict(skills.STACK_OPTIONS)), (skills.pose_skill(iiwa_fk), None)):
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
        ctrl.setup_problem_functions()
        dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, 0.0, Q, Y=Y)
        assert dq.shape == (B, 7) and np.array_equal(mode, rmode)
        assert pinv_close(dq, ref)
    qspec = skills.qp_skill(iiwa_fk)
    qctrl = cc.ReactiveQPController(skill_spec=qspec)
    qctrl.setup_problem_functions()
    qctrl.setup_solver()
    dq, _, slack, status = qctrl.solve_batch(0.0, Q, input_var=Y)
    rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(qspec, 0.0, Q, Y=Y)
    assert (status == 0).all() and (rstatus == 0).all() and slack.shape == (B, 6)
    assert _rel(dq, rdq).max() < 1e-8 and _rel(slack, rslack).max() < 1e-8


def test_rollout_with_time_trajectory(ur5_fk):
    """Rollout of a tracking skill whose target moves with time (ur5_moe2016_example2 cell 7):
    the per-tick time terms (values and exact derivatives for the feed-forward) are read in
    place from the device buffer; must equal the host loop with the same time stamps."""
    from casclik_amd import sym as cs
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    path = cs.vertcat(0.5 * cs.sin(0.1 * t) * cs.sin(0.1 * t) + 0.2, 0.5 * cs.cos(0.1 * t) + 0.25 * cs.sin(0.1 * t),
                      0.5 * cs.sin(0.1 * t) * cs.cos(0.1 * t) + 0.1)
    spec = cc.SkillSpecification("track", t, q, constraints=[
        cc.EqualityConstraint("move_point", p - path, gain=0.5, constraint_type="soft")])
    ctrl = _controller(spec)
    assert ctrl.kernel_name != "dynamic"
    rng = np.random.default_rng(8)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.1, size=(70, 6))
    dt, n_ticks = 0.05, 9
    times = 3.0 + dt * np.arange(n_ticks)
    qh = Q.copy()
    for tv in times:
        dq, _, _ = ctrl.solve_batch(float(tv), qh)
        qh = qh + dq * dt
    q_dev, dq_dev, _ = ctrl.rollout_batch(times, Q, dt=dt)
    assert np.abs(q_dev - qh).max() < 1e-10 and np.abs(dq_dev - dq).max() < 1e-9


@pytest.mark.parametrize("multidim", [False, True])
def test_moe2016_box_skill_on_the_denavit_hartenberg_chain(ur5_fk, multidim):
    """ur5_moe2016_example2.ipynb cells 2-11 as written: the UR5 from its classic DH table
    (converter.from_denavit_hartenberg), wall avoidance as three 1-D sets (8 modes) or one 3-D set
    (multidim_sets), the moving tracking target, through both controllers; the notebook's loop
    (:537-545) as an on-device rollout from its UR5_home."""
    from oracle import clik_oracle
    from casclik_amd import converter
    pi = np.pi
    fk = converter.from_denavit_hartenberg(
        joint_angles=["s" for _ in range(6)], link_lengths=[0., -0.425, -0.392, 0., 0., 0.],
        link_offsets=[0.089, 0., 0., 0.109, 0.095, 0.082], link_twists=[pi / 2, 0., 0., pi / 2, -pi / 2, 0.],
        joint_names=ur5_fk["joint_names"], upper_limits=ur5_fk["upper"], lower_limits=ur5_fk["lower"])
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    T_fk = fk["T_fk"]
    p_fk = cs.Function("p_fk", [t, q], [T_fk(q)[:3, 3]])
    x_min, x_max, y_min, y_max, z_min, z_max = 0.1, 0.6, -0.5, 0.4, -0.3, 0.25
    omega = 0.1
    path_des = cs.vertcat(0.5 * cs.sin(omega * t) * cs.sin(omega * t) + 0.2,
                          0.5 * cs.cos(omega * t) + 0.25 * cs.sin(omega * t),
                          0.5 * cs.sin(omega * t) * cs.cos(omega * t) + 0.1)
    path_cnstr = cc.EqualityConstraint(label="move_point2", expression=p_fk(t, q) - path_des, priority=10,
                                       constraint_type="soft", gain=0.15)
    if multidim:
        cons = [cc.SetConstraint(label="colav_box", expression=p_fk(t, q), set_min=np.array([x_min, y_min, z_min]),
                                 set_max=np.array([x_max, y_max, z_max]), priority=7, constraint_type="hard", gain=5e2),
                path_cnstr]
        opts = {"multidim_sets": True}
    else:
        cons = [cc.SetConstraint(label="colav_x", expression=p_fk(t, q)[0], set_min=x_min, set_max=x_max, priority=8,
                                 constraint_type="hard", gain=5e2),
                cc.SetConstraint(label="colav_y", expression=p_fk(t, q)[1], set_min=y_min, set_max=y_max, priority=7,
                                 constraint_type="hard", gain=5e2),
                cc.SetConstraint(label="colav_z", expression=p_fk(t, q)[2], set_min=z_min, set_max=z_max, priority=9,
                                 constraint_type="hard", gain=5e2),
                path_cnstr]
        opts = None
    spec = cc.SkillSpecification(label="box_move", time_var=t, robot_var=q, robot_vel_var=dq, constraints=cons)
    assert [c.label for c in spec.constraints][:2] == (["colav_box", "move_point2"] if multidim else ["colav_y", "colav_x"])
    home = np.array([-(50.0 / 180.0) * pi, -(160.0 / 180.0) * pi, -(110.0 / 180.0) * pi, -(90.0 / 180.0) * pi,
                     -(90.0 / 180.0) * pi, 0.0])
    rng = np.random.default_rng(12)
    Q = home + rng.uniform(-1.0, 1.0, size=(400, 6))
    ctrl = _controller(spec, opts)
    seen = set()
    for tval in (0.0, 25.0):
        dqs, _, mode = ctrl.solve_batch(tval, Q)
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, tval, Q)
        assert np.array_equal(mode, rmode)
        assert pinv_close(dqs, ref)
        seen |= set(mode.tolist())
    assert len(seen) >= 2
    # the simulation loop of the notebook, 150 ticks from UR5_home
    dt, vmax, n_ticks = 0.008, pi / 5, 150
    ts = dt * np.arange(n_ticks)
    q_dev, _, mode_dev = ctrl.rollout_batch(ts, home[None, :].repeat(2, axis=0), dt=dt, max_speed=vmax)
    qh = home[None, :].copy()
    for k in range(n_ticks):
        r, rm = clik_oracle.pinv_solve_batch(spec, opts, ts[k], qh)
        qh = qh + np.clip(r, -vmax, vmax) * dt
    assert np.abs(q_dev[0] - qh[0]).max() < 1e-8 and mode_dev[0] == rm[0]
    qp = cc.ReactiveQPController(skill_spec=spec)
    qp.setup_problem_functions()
    qp.setup_solver()
    qdq, _, qsl, st = qp.solve_batch(3.0, Q)
    rdq, _, rsl, rst = clik_oracle.qp_solve_batch(spec, 3.0, Q)
    assert np.array_equal(st, rst)
    ok = rst == 0
    assert ok.sum() > 100 and _rel(qdq[ok], rdq[ok]).max() < 1e-8


@pytest.mark.parametrize("B", [16384, 131072])
def test_baseline_full_sizes(iiwa_fk, B):
    """BASELINE.json configs 3 and 5 at their full sizes (16384 per GPU; 131072 = the 8-GPU job on one
    device): the whole batch against the C oracle (OpenMP, fast enough), plus size-independent
    properties - instances are independent, so a permuted batch gives the permuted answer bit for bit,
    and the small-batch (multi-wave) and large-batch (one-wave) kernels agree on shared instances."""
    from oracle.c_oracle import CPinvOracle
    spec = skills.stack_skill(iiwa_fk)
    ctrl = _controller(spec, skills.STACK_OPTIONS)
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=0, distribution="mixed")
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    margin = np.full(B, np.inf)
    # (the C restatement reads the descriptor oracle/baseline_desc.py writes down without the product's front-end)
    ref, _, ref_mode = CPinvOracle(None, skills.STACK_OPTIONS, baseline=("iiwa", "stack")).solve_batch(
        0.0, Q, Y=Y, margins_out=margin)
    # a mode decided by a tangent-cone value within rounding of its threshold could differ between two correct
    # evaluations (the kernels use push-through / Woodbury forms, not the literal order).  How close these inputs
    # get: the smallest distance of any decision of any instance's mode scan from flipping (pseudo_inverse.py:
    # 222-252 thresholds; 2e-6 at 131072 instances) against the ~1e-9 two evaluations of de differ by - so equal
    # modes are expected here, not luck; decisions planted within 1e-12 ... 1e-6 of a limit are pinned to the
    # reference's own run in test_gpu_refpins.py (iiwa_stack_boundary)
    print("smallest tangent-cone decision margin over %d instances: %.3e" % (B, margin.min()))
    assert margin.min() > 1e-7
    assert np.array_equal(mode, ref_mode)
    assert len(np.unique(mode)) == 2 and pinv_close(dq, ref)
    perm = np.random.default_rng(1).permutation(B)
    dq_p, _, mode_p = ctrl.solve_batch(0.0, Q[perm], input_var=Y[perm])
    assert np.array_equal(dq_p, dq[perm]) and np.array_equal(mode_p, mode[perm])
    # the first 4096 instances alone run on the multi-wave kernel; inside the big batch on whichever
    # kernel the size selects
    dq_s, _, mode_s = ctrl.solve_batch(0.0, Q[:4096], input_var=Y[:4096])
    assert np.array_equal(mode_s, mode[:4096]) and _rel(dq_s, dq[:4096]).max() < 1e-9
