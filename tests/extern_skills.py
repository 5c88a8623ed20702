"""Skills whose constraints lie outside the affine-in-features row table (they run as generated
device code, casclik_amd/codegen.py); shared by the CPU code-generation tests and the GPU parity tests."""
import casclik_amd as cc
from casclik_amd import sym as cs


def double_pendulum_skill(track=False):
    """double_pendulum_2D_comparison_of_controllers.ipynb cells 3-11 (and 31-33 with track=True)."""
    l_1, l_2 = 1.0, 0.75
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 2), cs.MX.sym("dq", 2)
    p_mid = cs.vertcat(l_1 * cs.cos(q[0]), l_1 * cs.sin(q[0]))
    p = cs.vertcat(l_1 * cs.cos(q[0]) + l_2 * cs.cos(q[0] + q[1]),
                   l_1 * cs.sin(q[0]) + l_2 * cs.sin(q[0] + q[1]))
    if track:
        p_des = 0.25 * cs.vertcat(cs.cos(0.5 * t), cs.sin(0.5 * t)) + cs.vertcat(1.0, 1.0)
    else:
        p_des = cs.vertcat(0.75, 0.5)
    table_height = -0.05
    cn = [cc.EqualityConstraint(label="min_dist_cnstr", expression=p_des - p, gain=1.0, constraint_type="soft"),
          cc.VelocitySetConstraint(label="speed_limit_cnstr", expression=q, set_min=-cs.vertcat(0.5, 0.5),
                                   set_max=cs.vertcat(0.5, 0.5)),
          cc.SetConstraint(label="table_midpoint_cnstr", expression=p_mid[1] - table_height, set_min=0.0,
                           set_max=cs.inf),
          cc.SetConstraint(label="table_endpoint_cnstr", expression=p[1] - table_height, set_min=0.0,
                           set_max=cs.inf)]
    return cc.SkillSpecification(label="move_to_point_skill", time_var=t, robot_var=q, robot_vel_var=dq,
                                 constraints=cn)



def moe_box_skill(fk, soft_walls=False):
    """The 'singular' QP skill of ur5_moe2016_example2.ipynb cells 6-8: three 1-D box sets on the tool position
    (hard, gain 5e2), a soft tracking equality on a time trajectory, hard joint-speed limits (12 rows, 9 variables).
    Returns (spec, home)."""
    import numpy as np
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    omega = 0.1
    path = cs.vertcat(0.5 * cs.sin(omega * t) * cs.sin(omega * t) + 0.2,
                      0.5 * cs.cos(omega * t) + 0.25 * cs.sin(omega * t),
                      0.5 * cs.sin(omega * t) * cs.cos(omega * t) + 0.1)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    p_home = fk["chain"].fk_numeric(home)[:3, 3]
    box = [(c - 0.15, c + 0.15) for c in p_home]       # the notebook also starts inside its box
    kw = dict(constraint_type="soft") if soft_walls else {}
    cons = [cc.SetConstraint("colav_%d" % i, p[i], set_min=lo, set_max=hi, priority=7 + i, gain=5e2, **kw)
            for i, (lo, hi) in enumerate(box)]
    cons.append(cc.EqualityConstraint("move_point", p - path, priority=10, constraint_type="soft", gain=0.15))
    cons.append(cc.VelocitySetConstraint("speed", q, set_min=-np.full(6, np.pi / 5), set_max=np.full(6, np.pi / 5),
                                         priority=0))
    return cc.SkillSpecification("box_move", t, q, constraints=cons), home


def mixed_frame_skill(fk):
    """7-DoF arm: products / functions of tool-frame entries, a virtual variable, input and time terms."""
    t, q, x, y = cs.MX.sym("t"), cs.MX.sym("q", 7), cs.MX.sym("x", 1), cs.MX.sym("y", 3)
    T = fk["T_fk"](q)
    p = T[:3, 3]
    c = cs.vertcat(0.4, 0.1, 0.5)
    cn = [cc.SetConstraint("sphere", cs.dot(p - c, p - c), set_min=0.25, set_max=1e10, priority=0),
          cc.EqualityConstraint("mix", cs.vertcat(T[0, 3] * T[1, 3] - y[0] * cs.sin(t),
                                                  T[2, 2] * q[1] + cs.exp(-x[0]) * y[1],
                                                  cs.sqrt(1.5 + T[0, 0] * T[1, 1]) - cs.cos(x[0] + 0.3 * t) / (2.0 + y[2] ** 2)),
                                gain=2.0, priority=1, constraint_type="soft"),
          cc.EqualityConstraint("rest", q - 0.1, gain=0.5, priority=2, constraint_type="soft")]
    return cc.SkillSpecification("mixed", t, q, virtual_var=x, input_var=y, constraints=cn)


def dual_quaternion_skill(fk, which="Q_dist2", for_pinv=False):
    """ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 3, 14-18 (and the comparison notebook's
    cells 37-38): the tool frame as a dual quaternion Q_fk(q), its deviation from a desired frame as an
    8-row soft EqualityConstraint, joint limits and a joint speed limit."""
    import numpy as np
    from casclik_amd import numpy_geom, casadi_geom
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    Q_fk = fk["dual_quaternion_fk"]
    q1, q2 = cs.SX.sym("q1", 8), cs.SX.sym("q2", 8)
    dual_quaternion_product = cs.Function("dualquatprod", [q1, q2], [casadi_geom.dual_quaternion_product(q1, q2)])
    dual_quaternion_conj = cs.Function("dualquatconj", [q1], [casadi_geom.dual_quaternion_conj(q1)])
    rpy = [5.0 * (np.pi / 180.0), 0.0, 0.0]
    xyz = [0.5, 0.0, 0.5]
    Q_des = numpy_geom.dual_quaternion_revolute(xyz, rpy, [1, 0, 0], 0.0)
    Q_id = numpy_geom.dual_quaternion_revolute([0., 0., 0.], [0., 0., 0.], [1., 0., 0.], 0.0)
    gain, prio = 10.0, 301
    if which in ("cart_dist", "quat_dist"):
        # ur5_dual_quaternion_comparison_of_controllers.ipynb cell 12: position read off the dual
        # quaternion, and the "improper" quaternion distance to p_des
        p1, p2 = cs.SX.sym("p1", 4), cs.SX.sym("p2", 4)
        quaternion_product = cs.Function("quatprod", [p1, p2], [casadi_geom.quaternion_product(p1, p2)])
        quaternion_conj = cs.Function("quatconj", [p1], [casadi_geom.quaternion_conj(p1)])
        p_des_a = np.hstack([np.array([0.5, 0.5, 0.5]), 0.0])
        Q_r = Q_fk(q)[:4]
        Q_d = Q_fk(q)[4:8]
        if which == "cart_dist":
            expr = 2 * quaternion_product(Q_d, quaternion_conj(Q_r))[:3]
        else:
            expr = Q_d[:3] - 0.5 * quaternion_product(p_des_a, Q_r)[:3]
        gain, prio = 1.0, 300
    elif which == "Q_dist1":
        expr = dual_quaternion_product(Q_fk(q), dual_quaternion_conj(Q_des)) - Q_id
    else:
        # conj(Q_des - Q_fk(q)) (x) Q_des: the comparison notebook writes it with the product (cell 34), the
        # other one with the right-multiplication matrix of Q_des and diag(-1,-1,-1,1,-1,-1,-1,1) (cell 18) -
        # the same eight expressions
        expr = dual_quaternion_product(dual_quaternion_conj(Q_des - Q_fk(q)), Q_des)
    dist = cc.EqualityConstraint(label=which + "_cnstr", expression=expr, constraint_type="soft", gain=gain,
                                 priority=prio)
    q_min, q_max = np.array(fk["lower"]), np.array(fk["upper"])
    max_speed = np.pi / 5
    if for_pinv:
        limits = [cc.SetConstraint(label="limit_q_" + str(i), expression=q[i], set_min=q_min[i], set_max=q_max[i],
                                   priority=i) for i in range(6)]
        cn = [dist] + limits
    else:
        cn = [dist,
              cc.SetConstraint(label="Joint_Limits", expression=q, set_min=q_min, set_max=q_max),
              cc.VelocitySetConstraint(label="Joint_speed_limits", expression=q,
                                       set_min=-cs.vertcat([max_speed] * 6), set_max=cs.vertcat([max_speed] * 6))]
    return cc.SkillSpecification(label=which, time_var=t, robot_var=q, robot_vel_var=dq, constraints=cn)


def random_expression(rng, leaves, depth, angles=False):
    """Random smooth expression over the leaves (division and sqrt guarded away from singularities).  `angles`: also the
    angle-type and saturating functions of round 4 (off by default: the regression seeds of tools/fuzz_*.py replay the
    random stream of the original ten operations)."""
    if depth == 0 or rng.random() < 0.15:
        leaf = leaves[int(rng.integers(len(leaves)))]
        return leaf if rng.random() < 0.8 else leaf * float(rng.uniform(-2.0, 2.0))
    op = int(rng.integers(15 if angles else 10))
    a = random_expression(rng, leaves, depth - 1, angles)
    if op >= 10:
        # (guarded inside their domains)
        b = random_expression(rng, leaves, depth - 1, angles)
        return [cs.atan2(a, 1.5 + cs.cos(b)), cs.tanh(a), cs.fmin(a, cs.fmax(b, -0.5)),
                cs.asin(0.9 * cs.sin(a)) + cs.acos(0.8 * cs.cos(b)), cs.atan(a * b)][op - 10]
    if op <= 3:
        b = random_expression(rng, leaves, depth - 1, angles)
        return [a + b, a - b, a * b, a / (2.5 + cs.sin(b))][op]
    if op == 4:
        return cs.sin(a)
    if op == 5:
        return cs.cos(a)
    if op == 6:
        return cs.sqrt(1.0 + a * a)
    if op == 7:
        return cs.exp(-(a * a) / (1.0 + a * a))
    if op == 8:
        return a ** int(rng.integers(2, 4))
    return cs.norm_2(cs.vertcat(a, 0.7, random_expression(rng, leaves, depth - 1, angles)))


