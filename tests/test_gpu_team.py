"""GPU parity of the four-lanes-per-instance ("team") kernel of the config-3 family
(casclik_amd/csrc/clik_pinv_team.hpp) against the CPU oracles and against the
lane-per-instance kernels, through the C ABI (CLIK_LANES selects the variant when the
handle is created)."""
import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs
from casclik_amd.constraints import EqualityConstraint, SetConstraint
from casclik_amd.skill_specification import SkillSpecification
from tolerances import PINV_RTOL, pinv_close, qp_close

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))


def _controller(spec, options, lanes, monkeypatch):
    monkeypatch.setenv("CLIK_LANES", str(lanes))
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(options))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    return ctrl


@pytest.mark.parametrize("robot", ["iiwa", "ur5"])
@pytest.mark.parametrize("dist", ["interior", "mixed"])
@pytest.mark.parametrize("B", [1, 37, 200, 1024])
def test_team_kernel_vs_numpy_oracle(iiwa_fk, ur5_fk, robot, dist, B, monkeypatch):
    from oracle import clik_oracle
    fk = iiwa_fk if robot == "iiwa" else ur5_fk
    spec = skills.stack_skill(fk)
    ctrl = _controller(spec, skills.STACK_OPTIONS, 4, monkeypatch)
    assert ("/team4" in ctrl.kernel_variant(B))
    Q, Y = skills.synthetic_inputs(fk, B, seed=11 + B, distribution=dist)
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, ref_mode = clik_oracle.pinv_solve_batch(spec, skills.STACK_OPTIONS, 0.0, Q, Y=Y)
    assert np.array_equal(mode, ref_mode)
    assert pinv_close(dq, ref)
    if dist == "mixed" and B >= 200:
        assert (mode == 0).any() and (mode == 1).any()     # both candidates are exercised


def test_team_kernel_vs_c_oracle_and_lane_kernel_full_size(iiwa_fk, monkeypatch):
    """BASELINE config 3 at its full size: the whole batch against the C oracle, and against the
    lane-per-instance kernel (same modes; velocities within the rounding of the two evaluation orders)."""
    from oracle.c_oracle import CPinvOracle
    spec = skills.stack_skill(iiwa_fk)
    B = 16384
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=0, distribution="mixed")
    team = _controller(spec, skills.STACK_OPTIONS, 4, monkeypatch)
    lane = _controller(spec, skills.STACK_OPTIONS, 1, monkeypatch)
    assert ("/team4" in team.kernel_variant(B)) and not ("/team4" in lane.kernel_variant(B))
    dq_t, _, mode_t = team.solve_batch(0.0, Q, input_var=Y)
    dq_l, _, mode_l = lane.solve_batch(0.0, Q, input_var=Y)
    margin = np.full(B, np.inf)
    ref, _, ref_mode = CPinvOracle(None, skills.STACK_OPTIONS, baseline=("iiwa", "stack")).solve_batch(
        0.0, Q, Y=Y, margins_out=margin)
    assert margin.min() > 1e-7          # (no tangent-cone decision of this batch is within rounding of flipping)
    assert np.array_equal(mode_t, ref_mode) and np.array_equal(mode_l, ref_mode)
    assert pinv_close(dq_t, ref)
    assert pinv_close(dq_t, dq_l)
    # a permuted batch gives the permuted answer bit for bit (no cross-instance coupling, no
    # dependence on the position inside the quad / wave / block)
    perm = np.random.default_rng(1).permutation(B)
    dq_p, _, mode_p = team.solve_batch(0.0, Q[perm], input_var=Y[perm])
    assert np.array_equal(dq_p, dq_t[perm]) and np.array_equal(mode_p, mode_t[perm])


def _custom_stack(fk, gain_matrix, one_sided):
    n = len(fk["joint_names"])
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    dq = cs.MX.sym("dq", n)
    y = cs.MX.sym("y", 7)
    T = fk["T_fk"](q)
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    if one_sided:
        limits = SetConstraint(label="joint_limits", expression=q, set_max=hi, priority=0)
    else:
        limits = SetConstraint(label="joint_limits", expression=q, set_min=lo, set_max=hi, priority=0)
    rng = np.random.default_rng(4)
    K = 10.0 * np.eye(6) + (rng.uniform(-1, 1, (6, 6)) if gain_matrix else 0.0)
    pose = EqualityConstraint(label="tool_pose", expression=skills._pose_expression(T, y),
                              gain=K if gain_matrix else 10.0, constraint_type="soft", priority=1)
    # joint-space task on a subset of the joints, time-dependent target (feedforward term)
    sub = cs.vertcat(q[0] - 0.3 * cs.sin(t), q[2] + 0.1, q[n - 1] - 0.2 * t)
    center = EqualityConstraint(label="some_joints", expression=sub, gain=1.5, constraint_type="soft", priority=2)
    return SkillSpecification(label="custom_stack", time_var=t, robot_var=q, robot_vel_var=dq, input_var=y,
                              constraints=[limits, pose, center])


@pytest.mark.parametrize("gain_matrix", [False, True])
@pytest.mark.parametrize("one_sided", [False, True])
@pytest.mark.parametrize("feedforward", [True, False])
def test_team_kernel_family_members(iiwa_fk, gain_matrix, one_sided, feedforward, monkeypatch):
    """Other members of the family through run-time instantiated kernels: matrix gain, one-sided
    limits, a joint-space task on three joints with time terms, feedforward on / off."""
    from oracle import clik_oracle
    spec = _custom_stack(iiwa_fk, gain_matrix, one_sided)
    opts = dict(skills.STACK_OPTIONS, feedforward=feedforward)
    ctrl = _controller(spec, opts, 4, monkeypatch)
    B = 300
    assert ("/team4" in ctrl.kernel_variant(B)), ctrl.kernel_variant(B)
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=21, distribution="mixed")
    for t in (0.0, 0.7):
        dq, _, mode = ctrl.solve_batch(t, Q, input_var=Y)
        ref, ref_mode = clik_oracle.pinv_solve_batch(spec, opts, t, Q, Y=Y)
        assert np.array_equal(mode, ref_mode)
        assert pinv_close(dq, ref)


def test_team_kernel_near_singular(iiwa_fk, monkeypatch):
    """Stretched-out arm (sigma_min^2 of the pose Jacobian below lam): the shifted factorisations
    A0 = JJ' + lam I, A1 = 2JJ' + lam I stay positive definite and agree with the oracle."""
    from oracle import clik_oracle
    spec = skills.stack_skill(iiwa_fk)
    ctrl = _controller(spec, skills.STACK_OPTIONS, 4, monkeypatch)
    rng = np.random.default_rng(8)
    B = 128
    Q = rng.normal(0.0, 1e-5, size=(B, 7))      # all joints near zero: the iiwa is singular there
    _, Y = skills.synthetic_inputs(iiwa_fk, B, seed=2)
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, ref_mode = clik_oracle.pinv_solve_batch(spec, skills.STACK_OPTIONS, 0.0, Q, Y=Y)
    assert np.array_equal(mode, ref_mode)
    assert np.isfinite(dq).all()
    # (near the singularity |dq| is ~1e3..1e5 and the problem is ill-conditioned by construction:
    # relative agreement of the whole vector at the documented level)
    assert _rel(dq, ref).max() < 1e-5


def test_default_selection_by_batch(iiwa_fk, monkeypatch):
    """Unset CLIK_LANES: the library picks the team kernel for small batches of the family and the
    lane-per-instance kernels above; a skill outside the family never gets it."""
    monkeypatch.delenv("CLIK_LANES", raising=False)
    ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(iiwa_fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    assert ("/team4" in ctrl.kernel_variant(16384))
    # beyond the team kernel's range: one lane per instance with the skill's numbers compiled in ("lanev") when the
    # value-specialised build is attached (the default where hipcc is available); without it one wave per mode up to
    # 32768 instances, one lane per instance above, the two-waves-per-SIMD build from 524288 on
    assert ctrl.value_kernel
    for B in (32768, 65536, 131072, 1 << 20):
        assert ctrl.kernel_variant(B).endswith("/lanev")
    img = cc.PseudoInverseController(skill_spec=skills.stack_skill(iiwa_fk),
                                     options=dict(skills.STACK_OPTIONS, function_opts={"jit_values": False}))
    img.setup_problem_functions()
    assert img.kernel_variant(16384).endswith("/team4") and img.kernel_variant(32768).endswith("/mp2")
    assert img.kernel_variant(131072).endswith("/lane")
    assert img.kernel_variant(1 << 20).endswith("/lane/occ2")      # (the ahead-of-time shapes' large-batch build)
    pose = cc.PseudoInverseController(skill_spec=skills.pose_skill(iiwa_fk))
    pose.setup_problem_functions()
    # (never the config-3 team kernel: a single-mode skill with forward kinematics runs four lanes per instance with the
    # sin / cos evaluations split over the quad up to 16384 instances, one lane per instance above)
    assert pose.kernel_variant(16384).endswith("/quadv" if pose.value_kernel else "/lane")
    assert pose.kernel_variant(16385).endswith("/lanev" if pose.value_kernel else "/lane")


@pytest.mark.parametrize("robot", ["iiwa", "ur5"])
def test_value_specialised_and_image_reading_team_kernels_agree(iiwa_fk, ur5_fk, robot, monkeypatch):
    """team4v (the skill's numbers compiled in, clik_pinv_attach_value_kernel) against team4 (the same kernel reading
    the skill image from memory): same modes; velocities equal up to the re-association the compiler may do with
    literal coefficients; both within tolerance of the oracle."""
    from oracle import clik_oracle
    fk = iiwa_fk if robot == "iiwa" else ur5_fk
    spec = skills.stack_skill(fk)
    monkeypatch.setenv("CLIK_JIT_VALUES", "1")
    val = _controller(spec, skills.STACK_OPTIONS, 4, monkeypatch)
    monkeypatch.setenv("CLIK_JIT_VALUES", "0")
    img = _controller(spec, skills.STACK_OPTIONS, 4, monkeypatch)
    B = 3000
    assert val.kernel_variant(B).endswith("/team4v") and img.kernel_variant(B).endswith("/team4")
    Q, Y = skills.synthetic_inputs(fk, B, seed=77, distribution="mixed")
    dq_v, _, mode_v = val.solve_batch(0.0, Q, input_var=Y)
    dq_i, _, mode_i = img.solve_batch(0.0, Q, input_var=Y)
    ref, ref_mode = clik_oracle.pinv_solve_batch(spec, skills.STACK_OPTIONS, 0.0, Q[:400], Y=Y[:400])
    assert np.array_equal(mode_v, mode_i) and np.array_equal(mode_v[:400], ref_mode)
    assert pinv_close(dq_v, dq_i) and pinv_close(dq_v[:400], ref)


def test_value_specialised_lane_kernel_of_the_stack_between_16385_and_32768(iiwa_fk):
    """the config-3 family beyond the team kernel's batch range: "lanev" (solo_tick with the numbers compiled in)
    against the oracle, the image-reading kernels and the team kernel on shared instances"""
    from oracle import c_oracle
    spec, opts = skills.stack_skill(iiwa_fk), dict(skills.STACK_OPTIONS)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
    ctrl.setup_problem_functions()
    plain = cc.PseudoInverseController(skill_spec=spec, options=dict(opts, function_opts={"jit_values": False}))
    plain.setup_problem_functions()
    B = 20000
    assert ctrl.kernel_variant(B).endswith("/lanev") and plain.kernel_variant(B).endswith("/mp2")
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=17, distribution="mixed")
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    dq2, _, mode2 = plain.solve_batch(0.0, Q, input_var=Y)
    co = c_oracle.CPinvOracle(spec, opts)
    ref, _, rmode = co.solve_batch(0.0, Q, Y=Y)
    assert set(np.unique(rmode)) == {0, 1}
    assert np.array_equal(mode, rmode) and np.array_equal(mode2, rmode)
    assert pinv_close(dq, ref) and pinv_close(dq, dq2)
    dq3, _, mode3 = ctrl.solve_batch(0.0, Q[:16384], input_var=Y[:16384])          # (team4v on the same instances)
    assert np.array_equal(mode3, rmode[:16384]) and pinv_close(dq3, dq[:16384])


@pytest.mark.parametrize("skill", ["pose", "position"])
def test_value_specialised_lane_kernel_of_single_mode_skills(iiwa_fk, skill):
    """BASELINE configs 1 / 2 (single task, one mode): the kernel is instantiated with the skill's numbers compiled in
    (no skill image, no LDS staging: "lanev") unless function_opts["jit_values"] = False; both match the oracle and
    each other"""
    from oracle import clik_oracle
    from casclik_amd import skills
    spec = skills.pose_skill(iiwa_fk) if skill == "pose" else skills.position_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    plain = cc.PseudoInverseController(skill_spec=spec, options={"function_opts": {"jit_values": False}})
    plain.setup_problem_functions()
    assert ctrl.kernel_variant(4096).endswith("quadv") and ctrl.kernel_variant(65536).endswith("lanev")
    assert plain.kernel_variant(4096).endswith("lane")
    Q, Y = skills.synthetic_inputs(iiwa_fk, 4133, seed=3, distribution="mixed")
    Y = Y[:, :spec.n_input_var]
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    from tolerances import pinv_close
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q[::7], Y=Y[::7])
    assert np.array_equal(mode[::7], rmode)
    assert pinv_close(dq[::7], ref)                 # (every instance held to ITS bound, 8 u kappa)
    dq2, _, mode2 = plain.solve_batch(0.0, Q, input_var=Y)
    # (two instantiations of the same source: equal to rounding - "mixed" batches of this size hold instances within 1e-4
    # of a singularity, where either kernel's rounding is amplified by up to 1e7: the rule's ceiling for all, 1e-13 for
    # the typical instance)
    assert np.array_equal(mode, mode2) and pinv_close(dq, dq2), np.abs(dq - dq2).max()
    assert np.median(np.abs(dq - dq2).max(axis=1)) < 1e-13


@pytest.mark.parametrize("skill", ["pose", "position"])
def test_four_lanes_per_instance_for_single_mode_skills(iiwa_fk, skill, monkeypatch):
    """BASELINE config 2 at its own batch sizes (4096 / 16384 instances = 64 / 256 waves of the lane kernel on 1024
    SIMDs): four lanes per instance, the sin / cos of the state variables split over the quad ("quadv",
    pinv_solve_static_values_quad_kernel) - the same velocities as one lane per instance (CLIK_QUAD_FRONT=0) to
    rounding, ragged batches included, and the oracle's within the stated rule (pseudo_inverse.py:259-451, one mode)"""
    from oracle import clik_oracle
    from tolerances import pinv_close
    spec = skills.pose_skill(iiwa_fk) if skill == "pose" else skills.position_skill(iiwa_fk)
    quad = cc.PseudoInverseController(skill_spec=spec)
    quad.setup_problem_functions()
    if not quad.value_kernel:
        pytest.skip("no value-specialised kernel attached (hipcc missing)")
    monkeypatch.setenv("CLIK_QUAD_FRONT", "0")
    lane = cc.PseudoInverseController(skill_spec=spec)
    lane.setup_problem_functions()
    for B in (1, 3, 63, 64, 65, 4096, 16384):
        assert quad.kernel_variant(B).endswith("/quadv") and lane.kernel_variant(B).endswith("/lanev")
        Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=40 + B % 7, distribution="mixed")
        Y = Y[:, :spec.n_input_var]
        a, _, ma = quad.solve_batch(0.0, Q, input_var=Y)
        b, _, mb = lane.solve_batch(0.0, Q, input_var=Y)
        # (two instantiations of the same source: equal to rounding, not necessarily to the bit - held to the rule's
        # ceiling at the default damping, 8 u (smax^2 + lam) / lam: "mixed" batches of this size hold instances within 1e-4 of
        # a singularity, where the rounding of either kernel is amplified by 1e7)
        assert np.array_equal(ma, mb) and pinv_close(a, b), (B, np.abs(a - b).max())
        assert np.median(np.abs(a - b).max(axis=1)) < 1e-13
        if B <= 4096:
            n = min(B, 300)
            ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q[:n], Y=Y[:n])
            assert np.array_equal(ma[:n], rmode) and pinv_close(a[:n], ref)
    # a joint angle beyond the fast sin / cos range in ONE lane of a quad takes the cold path there only
    Q, Y = skills.synthetic_inputs(iiwa_fk, 64, seed=2, distribution="interior")
    Y = Y[:, :spec.n_input_var]
    Q[5, 2] += 2.0 * np.pi * 40000.0
    Q[17, 6] -= 2.0 * np.pi * 30000.0
    a, _, _ = quad.solve_batch(0.0, Q, input_var=Y)
    b, _, _ = lane.solve_batch(0.0, Q, input_var=Y)
    assert pinv_close(a, b)
    ref, _ = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q, Y=Y)
    assert pinv_close(a, ref, ceiling=1e-6)


def test_resident_ticks_fed_from_outside(iiwa_fk):
    """clik_pinv_resident_run: ONE launch of the value-specialised team kernel runs four ticks, each on targets the host
    copies in behind a stream and publishes with a ticket (include/clik.h); every tick's velocities and modes equal
    those of an ordinary launch on the same inputs, the counters add up, and the kernel leaves by itself."""
    import time
    import torch
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    B, NT = 1000, 4
    if "team4v" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    Q, _ = skills.synthetic_inputs(iiwa_fk, B, seed=21, distribution="mixed")
    Ys = [skills.synthetic_inputs(iiwa_fk, B, seed=30 + k, distribution="mixed")[1] for k in range(NT)]
    Qd = torch.from_numpy(Q).cuda()
    want = [ctrl.solve_batch(0.0, Qd, input_var=torch.from_numpy(Yk).cuda()) for Yk in Ys]
    want_host = [(w[0].cpu(), w[2].cpu()) for w in want]
    Yd = torch.zeros((B, 7), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qd, Yd, NT, timeout_s=30.0)
    # (a stream of another priority: streams of one priority share a few hardware queues, and copies queued behind the
    # resident kernel would wait until its watchdog lets it go - seen in the long test run, never in this test alone)
    feed = ctrl.resident_feed_stream()       # (a stream that makes progress beside the resident kernel)
    waves = run["waves"]
    assert waves == ((B + 63) // 64) * 4
    try:
        for k in range(1, NT + 1):
            with torch.cuda.stream(feed):
                Yd.copy_(torch.from_numpy(Ys[k - 1]))                               # fresh targets ...
                run["ticket"][0:1].copy_(torch.tensor([k], dtype=torch.int32))     # ... then their ticket
            feed.synchronize()
            t0 = time.time()
            while True:
                with torch.cuda.stream(feed):
                    tk, dn = run["ticket"].cpu(), run["done"].cpu()
                if int(dn.min()) >= k or int(tk[32]) != 0 or time.time() - t0 > 25.0:
                    break
                time.sleep(0.001)
            assert int(tk[32]) == 0 and int(dn.min()) == k and int(dn.max()) == k, \
                "tick %d: ticket [in_seq, stop, waves, ticks_done, max_polls lo / hi, n_ticks, polls, wave] = %s, slots " \
                "min %d max %d after %.2f s" % (k, tk[[0, 32, 48, 49, 50, 51, 52, 53, 54]].tolist(), int(dn.min()),
                                                int(dn.max()), time.time() - t0)
            # (read back and compared on the HOST: a comparison kernel on the default stream can be queued behind the resident
            # kernel on the same hardware queue and then waits until the watchdog lets it go - which queue the default stream
            # shares depends on how many streams the process has made: seen in the long test run, never in this test alone)
            with torch.cuda.stream(feed):
                got, gmode = run["out"].cpu(), run["mode"].cpu()
            feed.synchronize()
            assert torch.equal(gmode, want_host[k - 1][1]) and torch.equal(got, want_host[k - 1][0]), k
    finally:
        with torch.cuda.stream(feed):
            run["ticket"][32:33].copy_(torch.tensor([1], dtype=torch.int32))      # (leave, whatever happened)
        feed.synchronize()
        run["stream"].synchronize()
    tk = run["ticket"].cpu()
    assert int(tk[48]) == waves and int(tk[49]) == NT


def test_resident_kernel_leaves_by_its_watchdog(iiwa_fk):
    """A resident kernel nobody feeds leaves by itself: its poll budget (timeout_s at a nominal 0.2 us per poll, clik.h)
    runs out, it writes stop = 2 and every wave returns - within the same order of magnitude as timeout_s (round 6: the
    earlier nominal figure of 2.5 us per poll made it a tenth of timeout_s)."""
    import time
    import torch
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    B = 256
    if "team4v" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=5, distribution="mixed")
    Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
    torch.cuda.synchronize()
    t0 = time.time()
    run = ctrl.resident_start(Qd, Yd, 3, timeout_s=0.5)
    with pytest.raises(cc.ResidentWatchdog) as err:      # (resident_wait: how a caller learns of it)
        ctrl.resident_wait(run)
    took = time.time() - t0
    assert "watchdog" in str(err.value) and "ticket 1" in str(err.value)
    tk = run["ticket"].cpu()
    assert int(tk[32]) == 2 and int(tk[49]) == 0, tk[[0, 32, 48, 49]].tolist()
    assert 0.1 < took < 5.0, took


def test_resident_ticks_of_a_single_mode_skill(iiwa_fk):
    """Round 5 (VERDICT r4 missing 3): resident ticks for the four-lanes-per-instance kernel of single-mode skills with
    forward kinematics (BASELINE config 2; pinv_resident_quad_kernel, one wave per 16 instances).  (a) ticks fed one by one
    from the host: targets copied in behind a stream, published with a ticket; every tick's velocities equal an ordinary
    launch on the same inputs.  (b) a ring of four slots with every ticket published ahead (the rows of tick k + 1
    requested before tick k's arithmetic): every output slot equals a launch on that slot's inputs.  (c) the state kept by
    the kernel is the config-3 family's only: refused."""
    import time
    import torch
    spec = skills.pose_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    B, NT = 1000, 4
    if "quadv" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised kernel attached (hipcc missing)")
    dev = lambda a: torch.from_numpy(a).cuda()          # noqa: E731
    Q, _ = skills.synthetic_inputs(iiwa_fk, B, seed=21, distribution="mixed")
    Ys = [skills.synthetic_inputs(iiwa_fk, B, seed=30 + k, distribution="mixed")[1] for k in range(NT)]
    Qd = dev(Q)
    want = [ctrl.solve_batch(0.0, Qd, input_var=dev(Yk)) for Yk in Ys]
    want_host = [(w[0].cpu(), w[2].cpu()) for w in want]
    Yd = torch.zeros((B, 7), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qd, Yd, NT, timeout_s=30.0)
    feed = ctrl.resident_feed_stream()       # (a stream that makes progress beside the resident kernel)
    assert run["waves"] == (B + 15) // 16
    try:
        for k in range(1, NT + 1):
            with torch.cuda.stream(feed):
                Yd.copy_(dev(Ys[k - 1]))
                run["ticket"][0:1].copy_(torch.tensor([k], dtype=torch.int32))
            feed.synchronize()
            t0 = time.time()
            while True:
                with torch.cuda.stream(feed):
                    tk, dn = run["ticket"].cpu(), run["done"].cpu()
                if int(dn.min()) >= k or int(tk[32]) != 0 or time.time() - t0 > 25.0:
                    break
                time.sleep(0.001)
            assert int(tk[32]) == 0 and int(dn.min()) == k and int(dn.max()) == k, (k, tk[[0, 32, 48, 49]].tolist())
            # (read back and compared on the HOST: a comparison kernel on the default stream can be queued behind the
            # resident kernel on the same hardware queue and would wait until its watchdog lets it go - seen in the long
            # test run, never in this test alone)
            with torch.cuda.stream(feed):
                got, gmode = run["out"].cpu(), run["mode"].cpu()
            feed.synchronize()
            assert torch.equal(gmode, want_host[k - 1][1]) and torch.equal(got, want_host[k - 1][0]), k
    finally:
        with torch.cuda.stream(feed):
            run["ticket"][32:33].copy_(torch.tensor([1], dtype=torch.int32))
        feed.synchronize()
        run["stream"].synchronize()
    assert int(run["ticket"].cpu()[49]) == NT
    # ---- (b) a ring of four slots, every ticket ahead
    D, NT = 4, 10
    batches = [skills.synthetic_inputs(iiwa_fk, B, seed=60 + k, distribution="mixed") for k in range(D)]
    wants = [ctrl.solve_batch(0.0, dev(q), input_var=dev(y)) for q, y in batches]
    Qr = torch.stack([dev(b[0]) for b in batches]).contiguous()
    Yr = torch.stack([dev(b[1]) for b in batches]).contiguous()
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qr, Yr, NT, timeout_s=20.0, ring_depth=D)
    feeder = ctrl.resident_feed(run, NT, closed_loop=False, timeout_s=20.0)
    run["stream"].synchronize()
    feeder.synchronize()
    tk = run["ticket"].cpu()
    assert int(tk[32]) == 0 and int(tk[49]) == NT and int(run["done"].min()) == NT
    for s_ in range(D):
        assert torch.equal(run["out"][s_], wants[s_][0]) and torch.equal(run["mode"][s_], wants[s_][2]), s_
    # ---- (c)
    with pytest.raises(Exception):
        ctrl.resident_start(Qd, Yd, 2, timeout_s=5.0, integrate_dt=1e-3, max_speed=2.0)


def test_resident_ticks_over_a_ring_of_input_slots(iiwa_fk):
    """Resident ticks with ticket->ring_depth = 4 (include/clik.h): tick k reads its rows from slot (k - 1) % 4 of the
    input rings and writes slot (k - 1) % 4 of the outputs.  (a) every ticket published ahead, DIFFERENT inputs in every
    slot, ten ticks: the software-pipelined path (rows of tick k + 1 requested in the middle of tick k) must pick the
    right slot each time - every output slot equals an ordinary launch on that slot's inputs.  (b) a producer on
    another stream refills the slot that has come free (every done[w] >= k + 1 - 4) with new inputs while the kernel
    runs and publishes tickets two ahead: twelve ticks, each tick's outputs equal those of a launch on ITS inputs."""
    import time
    import torch
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    B, D = 1000, 4
    if "team4v" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    batches = [skills.synthetic_inputs(iiwa_fk, B, seed=60 + k, distribution="mixed") for k in range(16)]
    dev = lambda a: torch.from_numpy(a).cuda()          # noqa: E731
    want = [ctrl.solve_batch(0.0, dev(q), input_var=dev(y)) for q, y in batches]

    # ---- (a) all tickets ahead
    NT = 10
    Qr = torch.stack([dev(batches[s][0]) for s in range(D)]).contiguous()
    Yr = torch.stack([dev(batches[s][1]) for s in range(D)]).contiguous()
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qr, Yr, NT, timeout_s=20.0, ring_depth=D)
    feeder = ctrl.resident_feed(run, NT, closed_loop=False, timeout_s=20.0)
    run["stream"].synchronize()
    feeder.synchronize()
    tk = run["ticket"].cpu()
    assert int(tk[32]) == 0 and int(tk[49]) == NT and int(run["done"].min()) == NT
    for s in range(D):
        assert torch.equal(run["out"][s], want[s][0]) and torch.equal(run["mode"][s], want[s][2]), s

    # ---- (b) a producer that refills slots while the kernel runs
    NT = 12
    Qr = torch.zeros((D, B, 7), dtype=torch.float64, device="cuda")
    Yr = torch.zeros((D, B, 7), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qr, Yr, NT, timeout_s=30.0, ring_depth=D)
    feed = ctrl.resident_feed_stream()       # (a stream that makes progress beside the resident kernel)
    seen = {}
    try:
        published = 0
        t0 = time.time()
        while len(seen) < NT and time.time() - t0 < 25.0:
            with torch.cuda.stream(feed):
                dn = int(run["done"].min().item())
            # collect the outputs of finished ticks before their slot is refilled
            for k in range(len(seen) + 1, dn + 1):
                with torch.cuda.stream(feed):
                    seen[k] = (run["out"][(k - 1) % D].clone(), run["mode"][(k - 1) % D].clone())
            # tick k may be published when its slot is free (tick k - D collected) and at most two ahead of `done`
            while published < NT and published - len(seen) < 2 and published + 1 - D <= len(seen):
                k = published + 1
                with torch.cuda.stream(feed):
                    Qr[(k - 1) % D].copy_(dev(batches[k][0]))
                    Yr[(k - 1) % D].copy_(dev(batches[k][1]))
                    run["ticket"][0:1].copy_(torch.tensor([k], dtype=torch.int32))
                feed.synchronize()
                published = k
        assert len(seen) == NT, (len(seen), run["ticket"].cpu()[[0, 32, 49]].tolist())
    finally:
        with torch.cuda.stream(feed):
            run["ticket"][32:33].copy_(torch.tensor([1], dtype=torch.int32))
        feed.synchronize()
        run["stream"].synchronize()
    for k in range(1, NT + 1):
        assert torch.equal(seen[k][0], want[k][0]) and torch.equal(seen[k][1], want[k][2]), k


def test_resident_ticks_integrate_the_state_and_take_streamed_targets(iiwa_fk):
    """ticket->integrate_dt > 0 (include/clik.h): the resident kernel reads q once and steps it itself,
    q += clamp(dq, +-max_speed) dt after every tick, while the TARGETS of every tick come from a ring that was filled
    ahead (what a producer of y does): 24 closed-loop ticks with a different target batch each, in one launch that
    never waits.  Every tick's clamped velocity and mode equal those of the host loop - one ordinary launch per tick
    on the state the previous ticks produced."""
    import torch
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    B, NT, dt, vmax = 1000, 24, 0.004, 1.5
    if "team4v" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    Q0, _ = skills.synthetic_inputs(iiwa_fk, B, seed=90, distribution="mixed")
    Ys = [skills.synthetic_inputs(iiwa_fk, B, seed=100 + k, distribution="interior")[1] for k in range(NT)]
    # the host loop (the notebooks'): solve -> clamp -> Euler
    q = torch.from_numpy(Q0).cuda()
    want = []
    for k in range(NT):
        dq, _, mode = ctrl.solve_batch(0.0, q, input_var=torch.from_numpy(Ys[k]).cuda())
        dq = dq.clamp(-vmax, vmax)
        want.append((dq.clone(), mode.clone()))
        q = q + dq * dt
    # one resident launch: the state in the kernel, the targets in a ring as deep as the run (all published ahead)
    Qr = torch.zeros((NT, B, 7), dtype=torch.float64, device="cuda")
    Qr[0] = torch.from_numpy(Q0).cuda()
    Yr = torch.stack([torch.from_numpy(y).cuda() for y in Ys]).contiguous()
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qr, Yr, NT, timeout_s=20.0, ring_depth=NT, integrate_dt=dt, max_speed=vmax)
    feeder = ctrl.resident_feed(run, NT, closed_loop=False, timeout_s=20.0)
    run["stream"].synchronize()
    feeder.synchronize()
    tk = run["ticket"].cpu()
    assert int(tk[32]) == 0 and int(tk[49]) == NT
    worst = 0.0
    for k in range(NT):
        assert torch.equal(run["mode"][k], want[k][1]), k
        worst = max(worst, float((run["out"][k] - want[k][0]).abs().max()))
    # (the same arithmetic in the same order: the kernel's fma(d, dt, q) against the host's q + dq * dt differ by
    # rounding in the state, which the ticks then see)
    assert worst < 1e-8, worst          # (measured 1.4e-9: the "mixed" inputs hold near-singular configurations)


def test_resident_ticks_with_input_rows_longer_than_eight(iiwa_fk):
    """A member of the family whose input_var has 14 entries (pose target + one target per joint): the resident kernel's
    quad hands its rows round two elements per lane and needs two rounds for such a row - ring of three slots with
    different batches, every ticket published ahead, seven ticks; every slot equals an ordinary launch."""
    import torch
    fk = iiwa_fk
    n = 7
    t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7 + n)
    T = fk["T_fk"](q)
    cons = [cc.EqualityConstraint("joints", cs.vertcat(*[q[j] - y[7 + j] for j in range(n)]), gain=1.0, priority=2),
            cc.SetConstraint(label="limits", expression=q, priority=0, set_min=np.array(fk["lower"]),
                             set_max=np.array(fk["upper"])),
            cc.EqualityConstraint("task", skills._pose_expression(T, y), gain=3.0, priority=1)]
    spec = cc.SkillSpecification("long_rows", t, q, input_var=y, constraints=cons)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options={"multidim_sets": True})
    ctrl.setup_problem_functions()
    B, D, NT = 600, 3, 7
    if "team4v" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    rng = np.random.default_rng(8)
    slots = []
    for s in range(D):
        Q, Y7 = skills.synthetic_inputs(fk, B, seed=200 + s, distribution="mixed")
        slots.append((Q, np.hstack([Y7, rng.uniform(-0.5, 0.5, size=(B, n))])))
    Qr = torch.stack([torch.from_numpy(a).cuda() for a, _ in slots]).contiguous()
    Yr = torch.stack([torch.from_numpy(b).cuda() for _, b in slots]).contiguous()
    want = [ctrl.solve_batch(0.0, Qr[s], input_var=Yr[s]) for s in range(D)]
    torch.cuda.synchronize()
    run = ctrl.resident_start(Qr, Yr, NT, timeout_s=20.0, ring_depth=D)
    feeder = ctrl.resident_feed(run, NT, closed_loop=False, timeout_s=20.0)
    run["stream"].synchronize()
    feeder.synchronize()
    tk = run["ticket"].cpu()
    assert int(tk[32]) == 0 and int(tk[49]) == NT
    for s in range(D):
        assert torch.equal(run["out"][s], want[s][0]) and torch.equal(run["mode"][s], want[s][2]), s


def test_resident_ticks_accept_only_what_fits_on_the_device_at_once(iiwa_fk):
    """every block of a resident launch must be running for a tick to complete, with room left for the ticket feeder: the
    launch wrapper admits one block per CU of this device.  On the MI355X: 16384 instances (256 blocks) run; 16448,
    32768 - which the hard-coded bound of round 3 let through, to spin until the watchdog - and anything larger are
    refused with CLIK_EUNSUPPORTED at once."""
    import torch
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    D, NT = 2, 4
    if "team4v" not in ctrl.kernel_variant(1000):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    dev = lambda a: torch.from_numpy(a).cuda()          # noqa: E731
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    fits = cus * 64
    for B in (fits, fits + 64, 2 * fits, 1 << 22):
        if B > fits:
            Qr = torch.zeros((B, 7), dtype=torch.float64, device="cuda")
            with pytest.raises(NotImplementedError) as err:
                ctrl.resident_start(Qr, Qr, 1, timeout_s=1.0)
            assert "resident" in str(err.value)
            continue
        batches = [skills.synthetic_inputs(iiwa_fk, B, seed=90 + k, distribution="mixed") for k in range(D)]
        want = [ctrl.solve_batch(0.0, dev(q), input_var=dev(y)) for q, y in batches]
        Qr = torch.stack([dev(q) for q, _ in batches]).contiguous()
        Yr = torch.stack([dev(y) for _, y in batches]).contiguous()
        torch.cuda.synchronize()
        run = ctrl.resident_start(Qr, Yr, NT, timeout_s=20.0, ring_depth=D)
        feeder = ctrl.resident_feed(run, NT, closed_loop=False, timeout_s=20.0)
        run["stream"].synchronize()
        feeder.synchronize()
        tk = run["ticket"].cpu()
        assert int(tk[32]) == 0 and int(run["done"].min()) == NT, (B, tk[[0, 32, 49]].tolist())
        for s in range(D):
            assert torch.allclose(run["out"][s], want[s][0], rtol=0, atol=1e-9) and torch.equal(run["mode"][s], want[s][2]), (B, s)


def test_resident_rows_through_lds_at_odd_and_tiny_batches(iiwa_fk):
    """Round 6: the config-3 resident kernel copies a wave's rows memory -> LDS without destination registers
    (stage_rows, clik_pinv_team.hpp) - in 16-byte pieces when base address and row pitch allow, in 4-byte pieces
    otherwise; a block that would run past the end of the ring is read from the ring's last sixteen rows.  Batches that
    take each path (odd: 56-byte pitch x odd row count; fewer than sixteen rows; ragged last wave; a misaligned base
    address), a ring of three slots with different inputs, every ticket ahead: every slot bit-equal to a launch."""
    import torch
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    if "team4v" not in ctrl.kernel_variant(1000):
        pytest.skip("no value-specialised team kernel attached (hipcc missing)")
    dev = lambda a: torch.from_numpy(a).cuda()          # noqa: E731
    D, NT = 3, 5
    for B, skew in ((333, 0), (5, 0), (1000, 0), (16, 0), (1000, 1)):
        batches = [skills.synthetic_inputs(iiwa_fk, B, seed=80 + k, distribution="mixed") for k in range(D)]
        want = [ctrl.solve_batch(0.0, dev(q), input_var=dev(y)) for q, y in batches]
        Qr = torch.stack([dev(q) for q, _ in batches]).contiguous()
        Yr = torch.stack([dev(y) for _, y in batches]).contiguous()
        if skew:        # (the same rows at an address that is 8 but not 16 bytes aligned)
            qb = torch.empty(Qr.numel() + 1, dtype=torch.float64, device="cuda")
            yb = torch.empty(Yr.numel() + 1, dtype=torch.float64, device="cuda")
            qb[1:].copy_(Qr.reshape(-1))
            yb[1:].copy_(Yr.reshape(-1))
            Qr, Yr = qb[1:].view(D, B, 7), yb[1:].view(D, B, 7)
            assert Qr.data_ptr() % 16 == 8
        torch.cuda.synchronize()
        run = ctrl.resident_start(Qr, Yr, NT, timeout_s=10.0, ring_depth=D)
        feeder = ctrl.resident_feed(run, NT, closed_loop=False, timeout_s=10.0)
        run["stream"].synchronize()
        feeder.synchronize()
        tk = run["ticket"].cpu()
        assert int(tk[32]) == 0 and int(run["done"].min()) == NT, (B, tk[[0, 32, 49]].tolist())
        for s in range(D):
            assert torch.equal(run["out"][s], want[s][0]) and torch.equal(run["mode"][s], want[s][2]), (B, skew, s)


def test_worst_case_of_the_team_sweeps_is_held_to_its_own_bound():
    """`tools/fuzz_team.py 20 3`, skill 15 - the largest error any randomised sweep of round 3 recorded for the pinv
    path, 2.32e-7 (profiles/r3_fuzz_summary.txt): iiwa, 6-row first task, damping 1e-9, mixed inputs.  Its projectors
    meet condition numbers of 3e9 ... 2e10, so the stated rule (tests/tolerances.py) bounds its instances by 3e-6 ... 2e-5
    - two orders above the default-options ceiling, which this skill was (rightly) not held to.  Every instance against
    ITS bound, both kernel variants."""
    import os
    import sys
    from oracle import clik_oracle
    from tolerances import rtol_from_cond, rel_err, ILL_POSED, U
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_team
    rng = np.random.default_rng(3)
    for _ in range(16):                      # replay the sweep's random stream up to skill 15
        d = fuzz_team.draw(rng)
    assert d["robot"] == "iiwa" and d["m"] == 6 and d["k3"] == 7 and 5e-10 < d["opts"]["damping_factor"] < 2e-9
    spec, opts, Q, Y, tval = d["spec"], d["opts"], d["Q"], d["Y"], d["tval"]
    kappa = np.zeros(len(Q))
    ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, tval, Q, Y=Y, cond_out=kappa)
    tol = rtol_from_cond(kappa)
    assert (tol < ILL_POSED).all() and tol.max() > 1e-6
    for values in ("1", "0"):
        os.environ["CLIK_JIT_VALUES"] = values
        try:
            ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
            ctrl.setup_problem_functions()
            dq, _, mode = ctrl.solve_batch(tval, Q, input_var=Y)
        finally:
            os.environ.pop("CLIK_JIT_VALUES", None)
        assert np.array_equal(mode, rmode)
        err = rel_err(dq, ref)
        assert (err <= tol).all(), (ctrl.kernel_variant(len(Q)), float((err / tol).max()))
        print("%s: worst err %.2e = %.2f u kappa = %.3f of its bound" % (
            ctrl.kernel_variant(len(Q)), err.max(), (err / (U * kappa)).max(), (err / tol).max()))


def test_lane_kernels_on_ragged_and_misaligned_batches(iiwa_fk):
    """The lane-per-instance value kernels (config 3 beyond 16384 instances, config 4 cold beyond 16384) on a batch whose
    last wave is ragged (20000 = 312 waves + 32 instances), with 16-byte aligned and with misaligned (8 mod 16) input
    arrays: bit-identical outputs, and the oracle's on a sample from both ends.  (Written for the round-6 experiment that
    moved the rows through LDS in 16-byte pieces - tools/experiments/lane_rows_lds.patch, retired - which got exactly
    these cases wrong.)"""
    import torch
    from oracle import clik_oracle
    B = 20000
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=17, distribution="mixed")

    def skewed(a):
        buf = torch.empty(a.size + 1, dtype=torch.float64, device="cuda")
        buf[1:].copy_(torch.from_numpy(a).reshape(-1))
        v = buf[1:].view(*a.shape)
        assert v.data_ptr() % 16 == 8
        return v
    # ---- config 3, lane kernel
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    if "lanev" not in ctrl.kernel_variant(B):
        pytest.skip("no value-specialised lane kernel attached (hipcc missing)")
    Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
    a = ctrl.solve_batch(0.0, Qd, input_var=Yd)
    b = ctrl.solve_batch(0.0, skewed(Q), input_var=skewed(Y))
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    idx = np.r_[0:64, B - 96:B]
    ref, rmode = clik_oracle.pinv_solve_batch(spec, ctrl.options, 0.0, Q[idx], Y=Y[idx])
    assert np.array_equal(a[2].cpu().numpy()[idx], rmode) and pinv_close(a[0].cpu().numpy()[idx], ref)
    # ---- config 4, lone-wave kernel (cold ticks beyond 16384 instances; the slack rows are six doubles wide)
    qspec = skills.qp_skill(iiwa_fk)
    qctrl = cc.ReactiveQPController(skill_spec=qspec)
    qctrl.setup_problem_functions()
    qctrl.setup_solver()
    if not qctrl.value_kernel:
        pytest.skip("no value-specialised QP kernel attached (hipcc missing)")
    qa = qctrl.solve_batch(0.0, Qd, input_var=Yd, use_hot=False)
    qb = qctrl.solve_batch(0.0, skewed(Q), input_var=skewed(Y), use_hot=False)
    assert torch.equal(qa[3], qb[3])
    for k in (0, 2):
        assert torch.allclose(qa[k], qb[k], rtol=0.0, atol=0.0, equal_nan=True), k
    rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(qspec, 0.0, Q[idx], Y=Y[idx])
    ok = rstatus == 0
    assert np.array_equal(qa[3].cpu().numpy()[idx] == 0, ok)
    assert qp_close(qa[0].cpu().numpy()[idx], rdq, rows=ok) and qp_close(qa[2].cpu().numpy()[idx], rslack, rows=ok)
