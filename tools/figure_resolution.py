#!/usr/bin/env python3
"""The resolution matrix of the reference-held pins: for every branch of SURVEY Appendix A / B and every quirk D1-D5, D12,
WHICH figure a notebook of the reference stores (real CasADi + qpOASES) rejects a controller that deviates there, and by how
many pixels - or that no stored output can tell.  Deviations are the `_wrong` switches of oracle/clik_oracle.py (test
infrastructure) and option / gain changes; every run goes through the notebook's own closed loop (tests/notebook_figures.py).

    python tools/figure_resolution.py [--quick]      -> profiles/r5_figure_resolution.{md,json}
(about 6 min on one core; --quick shortens the Moe runs to 1600 ticks as tests/test_figure_pins.py does)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import notebook_figures as cf                     # noqa: E402
from oracle import clik_oracle                    # noqa: E402

QUICK = "--quick" in sys.argv


# ---- runs --------------------------------------------------------------------------------------------------------------
def moe(case, wrong=None, options=None, mutate=None, n_ticks=None):
    fk = cf.moe_fk()
    kind, sit = case.split("_")
    spec = cf.moe_skill(fk, sit)
    if mutate:
        mutate(spec)
    opts = dict(cf.moe_options(case) or {})
    opts.update(options or {})
    if kind == "pinv":
        def solve(t, q):
            dz, mode = clik_oracle.pinv_solve_batch(spec, opts, float(t), q[None, :], _wrong=wrong)
            return dz[0], int(mode[0])
    else:
        def solve(t, q):
            dq, _, _, status = clik_oracle.qp_solve_batch(spec, float(t), q[None, :], _wrong=wrong, **(options or {}))
            return (dq[0] if status[0] == 0 else np.zeros(6)), None
    n = n_ticks or (1600 if QUICK else 2500)
    t_sim, _, p_sim, e_sim, mode_sim = cf.simulate_moe(solve, lambda q: fk["chain"].fk_numeric(q)[:3, 3], n)
    pins = [p for p in cf.moe_pins(case, t_sim, p_sim, e_sim, mode_sim) if p[2] > 0]
    worst = max(pins, key=lambda p: p[1])
    return float(worst[1]), worst[0]


def dqc(which, kind, wrong=None, options=None, n_ticks=4500):
    from casclik_amd import skills
    fk = skills.ur5()
    spec, error_norm = cf.dqc_skill(fk, which, kind)
    if kind == "pinv":
        def solve(t, q):
            return clik_oracle.pinv_solve_batch(spec, options, float(t), q[None, :], _wrong=wrong)[0][0]
    else:
        def solve(t, q):
            dq, _, _, status = clik_oracle.qp_solve_batch(spec, float(t), q[None, :], _wrong=wrong, **(options or {}))
            assert status[0] == 0
            return dq[0]
    t_sim, log_e = cf.simulate_dqc(error_norm, solve, n_ticks)
    pins = cf.dqc_pins(which, kind, t_sim, log_e)
    out = {}
    for key, worst, n, where in pins:
        out[key] = float(worst)
    return out


def cart(case, wrong=None, options=None):
    kind, spec, dt, p0, virt = cf.build(case)

    def solve(t, p, x):
        Q = np.array([[p]])
        X = None if x is None else np.array([[x]])
        if kind == "pinv":
            dz, _ = clik_oracle.pinv_solve_batch(spec, options, float(t), Q, X=X, _wrong=wrong)
            return dz[0, 0], (dz[0, 1] if virt else None)
        dq, dxv, _, status = clik_oracle.qp_solve_batch(spec, float(t), Q, X=X, _wrong=wrong, **(options or {}))
        return dq[0, 0], (dxv[0, 0] if virt else None)
    t_sim, p_sim, dp_sim = cf.simulate(case, solve)
    curves = ["dp"] if case.endswith("point") else ["p", "dp"]
    return max(cf.deviation_in_pixels(case, c, t_sim, p_sim if c == "p" else dp_sim)[0] for c in curves)


ROWS = []


def row(branch, where, deviation, figure, literal_px, wrong_px, verdict=None):
    if verdict is None:
        verdict = "REJECTED" if wrong_px is not None and wrong_px > max(2.0, 3.0 * literal_px) else \
            ("not resolved" if wrong_px is not None else "no stored output")
    ROWS.append(dict(branch=branch, reference=where, deviation=deviation, figure=figure,
                     literal_px=literal_px, deviating_px=wrong_px, verdict=verdict))
    print("%-38s %-30s literal %-6s deviating %-8s %s" % (branch, figure, literal_px, wrong_px, verdict), flush=True)


def main():
    t0 = time.time()
    # ---- D1: the doubly processed first equality (pseudo_inverse.py:317-326 + :382-396) -----------------------------
    lit = dqc("Q_dist2", "pinv")
    no_d1 = dqc("Q_dist2", "pinv", wrong="no_D1")
    row("D1 first equality processed twice", "pseudo_inverse.py:317-326,382-396", "processed once (textbook)",
        "dqc_Q_dist2_pinv (standstill in the UR5 home singularity)", round(lit["dqc_Q_dist2_pinv"], 2),
        round(no_d1["dqc_Q_dist2_pinv"], 1))
    row("D1 (same run, the band of all curves)", "pseudo_inverse.py:317-326,382-396", "processed once (textbook)",
        "dqc_Q_dist2_union", round(lit["dqc_Q_dist2_union"], 2), round(no_d1["dqc_Q_dist2_union"], 1))
    for which in ("cart_dist", "quat_dist", "Q_dist1"):
        a = dqc(which, "pinv")
        b = dqc(which, "pinv", wrong="no_D1")
        k = "dqc_%s_pinv" % which
        row("D1 away from the singularity", "pseudo_inverse.py:317-326,382-396", "processed once (textbook)", k,
            round(a[k], 2), round(b[k], 2))
    # ---- damping factor (pseudo_inverse.py:92-105; default 1e-7, :47) -------------------------------------------------
    for lam in (1e-9, 1e-5, 1e-3):
        d = dqc("Q_dist2", "pinv", options={"damping_factor": lam})
        row("damping factor 1e-7 (default)", "pseudo_inverse.py:47,92-105", "damping_factor = %g" % lam,
            "dqc_Q_dist2_pinv", round(lit["dqc_Q_dist2_pinv"], 2), round(d["dqc_Q_dist2_pinv"], 1))
    try:
        d = dqc("Q_dist2", "pinv", options={"pinv_method": "standard"}, n_ticks=600)["dqc_Q_dist2_pinv"]
        note = None
    except np.linalg.LinAlgError:
        d, note = None, "REJECTED: J J' is exactly singular at UR5_home - the undamped solve has no answer, the stored run exists"
    row("damped vs standard pinv", "pseudo_inverse.py:92-105", "pinv_method = standard", "dqc_Q_dist2_pinv",
        round(lit["dqc_Q_dist2_pinv"], 2), d if d is None else round(d, 1), verdict=note)
    # ---- the Moe-2016 runs: S, projection, mode order, cones, D2, feed-forward -----------------------------------------
    lit_m = moe("pinv_multidim")[0]
    lit_s = moe("pinv_singular")[0]
    w, k = moe("pinv_multidim", wrong="no_S")
    row("activation matrix S in rJ (a8)", "pseudo_inverse.py:289-298,352-355,401-404", "J instead of S J", k,
        round(lit_m, 2), round(w, 1))
    w, k = moe("pinv_singular", wrong="textbook_projection")
    row("N pinv(J) (Chiaverini)", "pseudo_inverse.py:387-394", "pinv(J N) (textbook)", k, round(lit_s, 2), round(w, 1))
    w, k = moe("pinv_multidim", wrong="active_first")
    row("mode order: fewest active sets first (a9)", "pseudo_inverse.py:107-130", "most active first", k,
        round(lit_m, 2), round(w, 1))
    w, k = moe("pinv_multidim", wrong="cone_1d_rows")
    row("multidim tangent cone (a7)", "pseudo_inverse.py:222-252", "row-wise 1-D rule (:162-185)", k, round(lit_m, 2),
        round(w, 1))
    w, k = moe("pinv_multidim", wrong="set_pushes_back")
    row("D2 active set adds rows, no velocity", "pseudo_inverse.py:398-405", "active set pushes back with its gain", k,
        round(lit_m, 2), round(w, 1))
    w, k = moe("pinv_singular", wrong="set_pushes_back")
    row("D2 (three 1-D walls)", "pseudo_inverse.py:398-405", "active set pushes back with its gain", k,
        round(lit_s, 2), round(w, 1))
    w, k = moe("pinv_singular", options={"feedforward": False})
    row("feed-forward term -Jt", "pseudo_inverse.py:320-321", "feedforward = False", k, round(lit_s, 2), round(w, 1))
    w, k = moe("pinv_multidim", wrong="boundary_flipped")
    row("D4 boundary counts as inside (1e-12)", "pseudo_inverse.py:174-185,224-228", "margins -1e-12", k,
        round(lit_m, 2), round(w, 2))
    w, k = moe("pinv_singular", wrong="boundary_flipped")
    row("D4 (three 1-D walls)", "pseudo_inverse.py:174-185", "margins -1e-12", k, round(lit_s, 2), round(w, 2))
    w, k = moe("pinv_multidim", wrong="multidim_loose")
    row("D5 multidim inside test is strict", "pseudo_inverse.py:224-228", "the 1-D function's loose test", k,
        round(lit_m, 2), round(w, 2))
    w, k = moe("pinv_multidim", wrong="no_D1")
    row("D1 on the Moe run (equality is LAST)", "pseudo_inverse.py:317-326,382-396", "processed once", k, round(lit_m, 2),
        round(w, 2))
    # ---- the 1-D cone on the cart (the rail end) -------------------------------------------------------------------------
    a = cart("pinv_traj")
    b = cart("pinv_traj", wrong="boundary_flipped")
    row("D4 1-D cone at the rail end", "pseudo_inverse.py:174-185", "margins -1e-12", "cart pinv_traj p / dp",
        round(a, 2), round(b, 2))
    b = cart("pinv_traj", wrong="active_first")
    row("mode order (1 set: 2 modes)", "pseudo_inverse.py:107-130", "active mode first", "cart pinv_traj p / dp",
        round(a, 2), round(b, 1))
    b = cart("pinv_traj", wrong="set_pushes_back")
    row("D2 (cart at the rail end)", "pseudo_inverse.py:398-405", "active set pushes back", "cart pinv_traj p / dp",
        round(a, 2), round(b, 1))
    # ---- QP --------------------------------------------------------------------------------------------------------------
    a = dqc("quat_dist", "qp")["dqc_quat_dist_qp"]
    b = dqc("quat_dist", "qp", options={"mu": 0.01})["dqc_quat_dist_qp"]
    row("weight shifter mu = 1e-3 (a11)", "reactive_qp.py:44,175-189", "mu = 1e-2", "dqc_quat_dist_qp (decay rate)",
        round(a, 2), round(b, 1))
    b = dqc("quat_dist", "qp", wrong="slack_times")["dqc_quat_dist_qp"]
    row("D12 slack weight mu + w", "reactive_qp.py:187 vs :331", "(1 + mu) w", "dqc_quat_dist_qp", round(a, 2), round(b, 2),
        verdict="unresolvable: every stored run has w = 1, where mu + w == (1 + mu) w bit for bit")
    a = cart("qp_traj")
    b = cart("qp_traj", options={"mu": 0.05})
    row("weight shifter mu (cart)", "reactive_qp.py:44", "mu = 5e-2", "cart qp_traj p / dp", round(a, 2), round(b, 1))
    # ---- nothing stored exercises these ---------------------------------------------------------------------------------
    row("D3 converge_final_set_to_max", "pseudo_inverse.py:337-379", "-", "-", None, None,
        verdict="unresolvable: no notebook sets the option (grep: 0 uses); pinned through the stand-in only")
    row("first VelocityEqualityConstraint", "pseudo_inverse.py:327-336,430-443", "-", "-", None, None,
        verdict="unresolvable: no notebook builds one (grep: 0 uses); pinned through the stand-in only")
    row("pinv_method = standard", "pseudo_inverse.py:103-104", "-", "-", None, None,
        verdict="no stored figure of a converged run (ur5_dual_quaternion_vs_transformation_matrix cell 24 sets damped + 1e-26); stand-in only")
    out = {"rows": ROWS, "quick": QUICK, "seconds": round(time.time() - t0, 1)}
    with open(os.path.join(ROOT, "profiles", "r5_figure_resolution.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "r5_figure_resolution.md"), "w") as f:
        f.write("# What the reference-held figures resolve (round 5)\n\n"
                "Every run: the notebook's own closed loop through `oracle/clik_oracle.py`, compared with the digitised\n"
                "pixels of the figure the notebook stores (real CasADi + qpOASES).  `literal` = the reference's algorithm as\n"
                "written, `deviating` = the same with ONE change.  Pixels; a pin passes below 1.0.\n"
                "Made by `python tools/figure_resolution.py%s` in %.0f s.\n\n" % (" --quick" if QUICK else "", time.time() - t0))
        f.write("| branch / quirk | reference | deviation | stored figure | literal px | deviating px | verdict |\n|---|---|---|---|---|---|---|\n")
        for r in ROWS:
            f.write("| %s | `%s` | %s | %s | %s | %s | %s |\n" % (r["branch"], r["reference"], r["deviation"], r["figure"],
                                                                 r["literal_px"], r["deviating_px"], r["verdict"]))


if __name__ == "__main__":
    main()
