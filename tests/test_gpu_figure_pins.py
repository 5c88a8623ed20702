"""The HIP controllers against the figures the reference's notebook stores (tests/test_figure_pins.py for the what and
how): the notebook's own loops - `solve(t_sim[i], p_sim[i][, x_sim[i]])[0].toarray()`, explicit Euler, 1200 ticks -
run on the device path through the reference-shaped API, and the simulated curves have to pass through the samples
digitised from the stored PNGs; they also equal the oracle's closed loop tick by tick."""
import numpy as np
import pytest

import casclik_amd as cc
import notebook_figures as cf
from test_figure_pins import PIXELS, curves_of, oracle_solver
from tolerances import pinv_close, qp_close

pytestmark = pytest.mark.gpu


def hip_solver(case):
    kind, spec, dt, p0, virt = cf.build(case)
    if kind == "qp":
        ctrl = cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0])       # cells 8, 33, 58
    else:
        ctrl = cc.PseudoInverseController(skill_spec=spec)                              # cells 19, 51, 76
    ctrl.setup_problem_functions()
    ctrl.setup_solver()

    def solve(t, p, x):
        res = ctrl.solve(t, p, x) if virt else ctrl.solve(t, p)
        return float(res[0].toarray()[0, 0]), (float(res[1].toarray()[0, 0]) if virt else None)
    return solve


@pytest.mark.parametrize("case", cf.CASES)
def test_hip_controllers_reproduce_the_figures_the_reference_stores(case):
    t_sim, p_sim, dp_sim = cf.simulate(case, hip_solver(case))
    for curve in curves_of(case):
        worst, n = cf.deviation_in_pixels(case, curve, t_sim, p_sim if curve == "p" else dp_sim)
        assert n > 60 and worst < PIXELS, (case, curve, worst)
    # ... and the oracle's closed loop, 1200 ticks long, is the same curve (the pinv runs switch modes at the rail
    # end: a tick earlier or later there would show as a speed-sized difference)
    _, p_ref, dp_ref = cf.simulate(case, oracle_solver(case))
    assert np.abs(p_sim - p_ref).max() < 1e-8 and np.abs(dp_sim - dp_ref).max() < 1e-7, (
        case, np.abs(p_sim - p_ref).max(), np.abs(dp_sim - dp_ref).max())


@pytest.mark.parametrize("case", cf.PENDULUM_CASES)
def test_hip_qp_reproduces_the_double_pendulum_figures(case):
    """double_pendulum_2D_comparison_of_controllers.ipynb cells 14-19 / 35-38: the ReactiveQPController with the table
    SetConstraints (general inequality rows: the mixed-family kernel) and saturating joint-speed limits"""
    from test_figure_pins import pendulum_deviations, pendulum_oracle_solver
    ctrl = cc.ReactiveQPController(skill_spec=cf.pendulum_skill(case), robot_var_weights=[1.0, 1.0])
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    t_sim, q_sim, dq_sim, p_sim = cf.simulate_pendulum(case, lambda t, q: ctrl.solve(t, q)[0].toarray()[:, 0])
    for fig, curve, worst, n in pendulum_deviations(case, t_sim, dq_sim, p_sim):
        assert n > 60 and worst < PIXELS, (fig, curve, worst, n)
    _, q_ref, dq_ref, _ = cf.simulate_pendulum(case, pendulum_oracle_solver(case))
    assert np.abs(q_sim - q_ref).max() < 1e-7 and np.abs(dq_sim - dq_ref).max() < 1e-6, (
        np.abs(q_sim - q_ref).max(), np.abs(dq_sim - dq_ref).max())


def test_hip_pinv_reproduces_the_ur5_figure(ur5_fk):
    """ur5_transformation_matrix_comparison_of_controllers.ipynb cells 27-32: PseudoInverseController, the UR5's tool to
    (0.5, 0.5, 0.5) from UR5_home through the notebook's own loop"""
    ctrl = cc.PseudoInverseController(skill_spec=cf.ur5_pinv_point_skill(ur5_fk))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    t_sim, p_sim = cf.simulate_ur5(ur5_fk, lambda t, q: ctrl.solve(t, q)[0].toarray()[:, 0])
    for k, curve in enumerate("xyz"):
        worst, n = cf.deviation_in_pixels("ur5_pinv_p", curve, t_sim, p_sim[:, k])
        assert n >= 15 and worst < PIXELS, (curve, worst, n)


@pytest.mark.parametrize("kind", ["qp", "pinv"])
def test_hip_controllers_retrace_the_ur5_point_frame_figures(ur5_fk, kind):
    """the same notebook's frame_3d figures, cells 17 (the ReactiveQPController's point run - three error rows, the
    multidimensional joint limits, joint-speed rows: the only figure that run has) and 33 (the pinv run of cell 32):
    tool path and frame-axis tips, 1 px = 6 mm, the stored image's crop offset read off the black dot at p_des"""
    if kind == "qp":
        ctrl = cc.ReactiveQPController(skill_spec=cf.ur5_qp_point_skill(ur5_fk))
    else:
        ctrl = cc.PseudoInverseController(skill_spec=cf.ur5_pinv_point_skill(ur5_fk))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    q_sim, dq_sim = cf.simulate_ur5_joints(lambda t, q: ctrl.solve(t, q)[0].toarray()[:, 0], clamp=(kind == "pinv"),
                                           return_dq=True)
    dev = cf.ur5_point_frame_pins(ur5_fk, kind, q_sim)
    for colour, (worst, covered, n) in dev.items():
        assert n > 50 and worst < cf.FRAME_PIXELS and covered > (0.97 if colour == "k" else 0.8), (kind, dev)
    if kind == "pinv":
        joints = cf.ur5_point_joint_pins(q_sim, dq_sim)      # (cell 31: joint positions and clamped joint speeds)
        for name, curves in joints.items():
            for worst, covered, n in curves:
                assert n >= 15 and worst < cf.FRAME_PIXELS and covered > 0.8, (name, joints)
        print("ur5 point pinv joints figure:", {k: [round(c[0], 2) for c in v] for k, v in joints.items()})
    print("ur5 point %s frame_3d figure: %s" % (kind, {c: (round(v[0], 2), round(v[1], 2)) for c, v in dev.items()}))


def test_hip_qp_reproduces_the_ur5_input_experiment_figure(ur5_fk):
    """ur5_input_experiment.ipynb cells 15-17 with the notebook's own calls: `setup_initial_problem_solver()`,
    `solve_initial_problem(time_var0=0, robot_var0=UR5_home, input_var0=[0, 0, 0])[-1]`, then 4500 ticks of
    `solve(t, q, input_var=y, warmstart_slack_var=slack)`"""
    from test_figure_pins import INPUT_PIXELS
    ctrl = cc.ReactiveQPController(skill_spec=cf.ur5_input_skill(ur5_fk))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    ctrl.setup_initial_problem_solver()
    state = {"slack": ctrl.solve_initial_problem(time_var0=0, robot_var0=cf.UR5_HOME, input_var0=[0, 0, 0])[-1]}

    def solve(t, q, y):
        res = ctrl.solve(t, q, input_var=y, warmstart_slack_var=state["slack"])
        if res[-1] is not None:
            state["slack"] = res[-1].toarray()[:, 0]
        return res[0].toarray()[:, 0]
    t_sim, p_sim = cf.simulate_ur5_input(ur5_fk, solve)
    for k, curve in enumerate("xyz"):
        worst, n = cf.deviation_in_pixels("ur5_qp_input", curve, t_sim, p_sim[:, k])
        assert n > 150 and worst < INPUT_PIXELS, (curve, worst, n)


@pytest.mark.parametrize("which,figure,above,least", [("Q_dist1", "ur5_qdist1_e", -13.0, 25), ("Q_dist2", None, 0, 0),
                                                       ("T_dist2", None, 0, 0)])
def test_hip_qp_reproduces_the_error_decay_of_the_frame_figures(ur5_fk, which, figure, above, least):
    """ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 24-27 (log axis) with the notebook's calls: the 8-row
    dual-quaternion deviation runs as generated code in the instantiated QP kernel; Q_dist2 / T_dist2: the runs'
    POSES against the stored frame_3d figures, cells 33 / 34"""
    spec, _, error_norm = cf.frame_error_skill(ur5_fk, which, "qp")
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    ctrl.setup_initial_problem_solver()
    state = {"slack": ctrl.solve_initial_problem(0, cf.UR5_HOME)[-1]}

    def solve(t, q):
        res = ctrl.solve(t, q, warmstart_slack_var=state["slack"])
        if res[-1] is not None:
            state["slack"] = res[-1].toarray()[:, 0]
        return res[0].toarray()[:, 0]
    t_sim, log_e, q_sim = cf.simulate_frame_error(error_norm, solve, return_q=True)
    if figure is not None:
        worst, n = cf.deviation_in_pixels(figure, "qp", t_sim, log_e, above=above)
        assert n >= least and worst < PIXELS, (worst, n)
    else:
        assert log_e[-1] < -13.0
        dev = cf.frame_error_frame_pins(ur5_fk, which, q_sim)
        for colour, (worst, covered, n) in dev.items():
            assert n > 100 and worst < cf.FRAME_PIXELS and covered > (0.97 if colour == "k" else 0.8), (which, dev)
        print("%s qp frame_3d figure: %s" % (which, {c: (round(v[0], 2), round(v[1], 2)) for c, v in dev.items()}))


@pytest.mark.parametrize("case", ["qp_point", "pinv_point", "qp_traj", "pinv_traj", "qp_path"])
def test_on_device_rollouts_reproduce_the_cart_figures(case):
    """The same stored figures through the ON-DEVICE loop (solve -> clamp -> integrate inside one launch,
    `rollout_batch`: SURVEY 8(f).1): 1200 ticks in launches of four, the speed of each launch's last tick and the
    position it ends on sampled against the digitised curves.  (The pinv runs saturate at 0.275 m/s as the notebook's
    loop does; the pinv PATH run also clamps the path speed, which the rollout leaves free - not run here.)"""
    kind, spec, dt, p0, virt = cf.build(case)
    ctrl = (cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0]) if kind == "qp"
            else cc.PseudoInverseController(skill_spec=spec))
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    chunk = 4
    p, x = np.array([[p0]]), np.array([[0.0]])
    ts, ps, dps = [], [], []
    for k in range(0, cf.N_TICKS - chunk, chunk):
        times = dt * np.arange(k, k + chunk)
        res = ctrl.rollout_batch(times, p, dt=dt, max_speed=cf.MAX_SPEED if kind == "pinv" else 0.0,
                                 virtual_var=x if virt else None)
        if virt:
            p, x, dp = res[0], res[1], res[2]
        else:
            p, dp = res[0], res[1]
        ts.append(dt * (k + chunk - 1))        # the launch's last tick: its speed ...
        dps.append(float(dp[0, 0]))
        ps.append(float(p[0, 0]))              # ... and the position it integrates to, one tick later
    ts, ps, dps = np.array(ts), np.array(ps), np.array(dps)
    worst, n = cf.deviation_in_pixels(case, "dp", ts, dps)
    assert n > 60 and worst < PIXELS, (case, "dp", worst)
    if not case.endswith("point"):
        worst, n = cf.deviation_in_pixels(case, "p", ts + dt, ps)
        assert n > 60 and worst < PIXELS, (case, "p", worst)


# ---- ur5_moe2016_example2.ipynb cells 2-27: pinv with an ACTIVE multidimensional set / 8-mode wall sets, QP -------------
def moe_hip_controller(case):
    fk = cf.moe_fk()
    kind, sit = case.split("_")
    spec = cf.moe_skill(fk, sit)
    if kind == "pinv":
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=cf.moe_options(case))            # cell 11
    else:
        ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    return fk, kind, ctrl


@pytest.mark.parametrize("case", cf.MOE_CASES)
def test_hip_controllers_reproduce_the_moe_2016_figures(case):
    """cell 12's loop with the notebook's own calls - `setup_initial_problem_solver()`, `solve_initial_problem(0,
    UR5_home)[-1]`, 10000 x `solve(t_sim[i], q_sim[i, :], warmstart_slack_var=slack_res)`, `current_mode` - on the
    device path; every pin of tests/test_figure_pins.py within a pixel, and the closed loop equal to the oracle's"""
    from test_figure_pins import assert_moe_pins, moe_oracle_run
    fk, kind, ctrl = moe_hip_controller(case)
    ctrl.setup_initial_problem_solver()
    state = {"slack": ctrl.solve_initial_problem(0, cf.MOE_HOME)[-1]}

    def solve(t, q):
        res = ctrl.solve(t, q, warmstart_slack_var=state["slack"])
        if res[-1] is not None:
            state["slack"] = res[-1].toarray()[:, 0]
        return res[0].toarray()[:, 0], (ctrl.current_mode if kind == "pinv" else None)
    t_sim, q_sim, p_sim, e_sim, mode_sim = cf.simulate_moe(solve, lambda q: fk["chain"].fk_numeric(q)[:3, 3])
    worst = assert_moe_pins(case, t_sim, p_sim, e_sim, mode_sim)
    print("%s through the HIP solve(): worst pin %.2f px" % (case, worst))
    if case == "pinv_singular":
        assert cf.fill_deviation("moe_modes_separate", t_sim, mode_sim) == (0, 19)
    ref = moe_oracle_run(case)
    if kind == "pinv":
        # 10000 ticks with ~15 (multidim) / ~540 (8 modes, chattering) mode switches: the same modes tick by tick
        # outside the chattering stretches, the same joint trajectory
        same = (mode_sim == ref[4]).mean()
        assert np.abs(q_sim - ref[1]).max() < 1e-5 and same > 0.995, (np.abs(q_sim - ref[1]).max(), same)
    else:
        assert np.abs(p_sim - ref[2]).max() < 1.5e-3, np.abs(p_sim - ref[2]).max()      # (walls with K dt = 4: see the CPU test)


@pytest.mark.parametrize("case", cf.MOE_CASES)
def test_on_device_rollouts_reproduce_the_moe_2016_figures(case):
    """the same four runs through the ON-DEVICE loop (`rollout_batch`: solve -> clamp at pi / 5 -> Euler step inside
    the kernel), 10000 ticks in launches of five; the state, the tracking error and the mode at the end of every launch
    against the same pins (a launch = 0.04 s = 0.12-0.8 pixel columns)"""
    fk, kind, ctrl = moe_hip_controller(case)
    chunk = 5
    q = cf.MOE_HOME[None, :].copy()
    ts, ps, es, ms = [0.0], [fk["chain"].fk_numeric(q[0])[:3, 3]], [None], [0.0]
    es[0] = float(np.linalg.norm(ps[0] - np.array(cf.moe_path(0.0))))
    for k in range(0, cf.MOE_TICKS, chunk):
        times = np.array([cf.MOE_DT * i for i in range(k, k + chunk)])
        res = ctrl.rollout_batch(times, q, dt=cf.MOE_DT, max_speed=figure_max_speed())
        q = res[0]
        if kind == "qp":
            assert res[-1][0] == 0
        p = fk["chain"].fk_numeric(q[0])[:3, 3]
        ts.append(cf.MOE_DT * (k + chunk))                    # q after tick k + chunk - 1 = q_sim[k + chunk]
        ps.append(p)
        es.append(float(np.linalg.norm(p - np.array(cf.moe_path(times[-1])))))     # (e_sim[i + 1] against the path at t_sim[i])
        ms.append(float(res[2][0]) if kind == "pinv" else 0.0)
    from test_figure_pins import assert_moe_pins
    worst = assert_moe_pins(case, np.array(ts), np.array(ps), np.array(es), np.array(ms))
    print("%s through the on-device rollout: worst pin %.2f px" % (case, worst))


def figure_max_speed():
    import figure_skills
    return figure_skills.MOE_MAX_SPEED


@pytest.mark.parametrize("which,kind", cf.DQC_CASES)
def test_hip_controllers_reproduce_the_dual_quaternion_comparison_figures(ur5_fk, which, kind):
    """ur5_dual_quaternion_comparison_of_controllers.ipynb cells 17 / 39 with the notebook's own calls, all eight runs
    (four dual-quaternion task errors x both controllers; the constraints run as generated device code): the error norm
    over thirteen decades against the stored log-axis figures, and against the oracle at every 150th state of the run"""
    from oracle import clik_oracle
    spec, error_norm = cf.dqc_skill(ur5_fk, which, kind)
    ctrl = cc.PseudoInverseController(skill_spec=spec) if kind == "pinv" else cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    ctrl.setup_initial_problem_solver()
    state = {"slack": ctrl.solve_initial_problem(0, cf.UR5_HOME)[-1], "i": 0, "worst": 0.0}

    def solve(t, q):
        res = ctrl.solve(t, q, warmstart_slack_var=state["slack"])
        if res[-1] is not None:
            state["slack"] = res[-1].toarray()[:, 0]
        dq = res[0].toarray()[:, 0]
        if state["i"] % 150 == 0:
            if kind == "pinv":
                ref = clik_oracle.pinv_solve_batch(spec, None, float(t), q[None, :])[0]
                assert pinv_close(dq[None, :], ref), (which, state["i"])
            else:
                ref = clik_oracle.qp_solve_batch(spec, float(t), q[None, :])[0]
                assert qp_close(dq[None, :], ref), (which, state["i"])
        state["i"] += 1
        return dq
    t_sim, log_e, q_sim = cf.simulate_dqc(error_norm, solve, return_q=True)
    pins = cf.dqc_pins(which, kind, t_sim, log_e)
    assert len(pins) == 2
    for key, worst, n, where in pins:
        assert n > 50 and worst < PIXELS, (key, worst, n, where)
    if (which, kind) != ("Q_dist2", "pinv"):
        # ... and the run's POSES against the stored frame_3d figure (cells 22-27, 44-48: the tool's path and the tips of
        # its frame's axes, matplotlib's projection restated in tests/golden/frame3d_pins.py).  PINV(Q_dist2) leaves the
        # home singularity through saturated velocities - its stored 3-D figure is one member of a family a 1e-9 rad
        # change of the start spreads out, see frame3d_pins.DQC_FRAMES - and keeps the error-norm pin only.
        dev = cf.dqc_frame_pins(ur5_fk, which, kind, q_sim)
        for colour, (worst, covered, n) in dev.items():
            assert n > 250 and worst < cf.FRAME_PIXELS and covered > (0.97 if colour == "k" else 0.8), (which, kind, dev)
        print("%s %s frame_3d figure: %s" % (which, kind, {c: (round(v[0], 2), round(v[1], 2)) for c, v in dev.items()}))
    if (which, kind) == ("cart_dist", "pinv"):
        band, px = cf.FIGS["dqc_cart_dist_pinv_band"], cf.FIGS["dqc_cart_dist_pinv_pixel"][1]
        tail = log_e[t_sim > 12.0]          # (the chatter on the unreachable target: the stored band's amplitude)
        assert abs(tail.min() - band[0]) < 1.5 * px and abs(tail.max() - band[1]) < 1.5 * px
    print("%s %s through the HIP solve(): %s" % (which, kind, [(k, round(w, 2)) for k, w, _, _ in pins]))
