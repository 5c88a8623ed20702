#!/usr/bin/env python3
"""Which four STARTS of the box QP's active-set passes should the four waves of a block run (clik_qp_static.hpp, FOLIO)?
Numpy model of qp_box_pas on the bench inputs of BASELINE config 4 (the functions of tools/qp_portfolio_study.py): pass
counts per instance for forward / reverse / over-relaxed (1.5) / symmetric Gauss-Seidel starts of 6, 12 and 18 sweeps, a
time model (0.125 us per sweep, 0.95 us per pass: profiles/r3_qp_portfolio_study.md) and, for every set of four, the
slowest instance's time when each instance is done as soon as ANY of the four has finished it.
    python tools/qp_wave_portfolio_study.py [instances = 16384] [seed = 0]        (about five minutes)
Result on 16384 instances, seed 0: the lone start (forward x 12) 6.25 us; best sets of four 4.55 us, among them
(forward x 6, forward x 12, reverse x 6, relaxed x 18) - round 4's first choice; best pair (reverse x 6, relaxed x 18)
5.10 us.  Over seeds 0 - 3 (and with reverse relaxed sweeps and 3 / 9 / 24 sweeps in the pool) the set with the shortest
slowest instance on average is (forward x 3, reverse x 6, relaxed x 12, reverse relaxed x 3): 4.55 / 4.18 / 5.12 / 4.35 us
against 4.55 / 4.55 / 6.45 / 5.3 of the first choice and 6.25 / 7.2 / 7.2 / 7.2 of the lone start - the shipped one."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.qp_pass_study import box_qps              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
P, g, lb, ub = box_qps(B, seed=SEED)
src = open(os.path.join(ROOT, "tools", "qp_portfolio_study.py")).read().split("# optimal partition via long PAS")[0]
exec(src.replace("P, g, lb, ub = box_qps(B)", "pass"))      # gs(), pas(), face_solve() as that study defines them
T_SWEEP, T_PASS = 0.125, 0.95


def start(kind, sweeps):
    if kind == "fwd":
        return gs(P, g, lb, ub, sweeps)                                      # noqa: F821
    if kind == "rev":
        return gs(P, g, lb, ub, sweeps, order=list(range(6, -1, -1)))        # noqa: F821
    if kind == "sor":
        return gs(P, g, lb, ub, sweeps, omega=1.5)                           # noqa: F821
    x = None
    for _ in range(sweeps // 2):
        x = gs(P, g, lb, ub, 1, order=list(range(7)), x0=x)                  # noqa: F821
        x = gs(P, g, lb, ub, 1, order=list(range(6, -1, -1)), x0=x)          # noqa: F821
    return x


res = {}
for kind in ("fwd", "rev", "sor", "sym"):
    for sw in (6, 12, 18):
        xs = start(kind, sw)
        its, _ = pas(P, g, lb, ub, xs, (xs <= lb) | (xs >= ub), "worst")     # noqa: F821
        res[(kind, sw)] = its
        print("%-4s x %2d  passes mean %.3f worst %d   slowest instance %.2f us" % (kind, sw, its.mean(), its.max(),
                                                                                  sw * T_SWEEP + its.max() * T_PASS))
time_of = {k: v * T_PASS + k[1] * T_SWEEP for k, v in res.items()}
for size in (2, 4):
    best = sorted(((np.minimum.reduce([time_of[k] for k in c]).max(), c) for c in itertools.combinations(time_of, size)),
                  key=lambda b: b[0])
    print("best sets of %d:" % size)
    for t, c in best[:6]:
        print("   %.2f us  %s" % (t, c))
print("the lone start (fwd x 12): %.2f us" % time_of[("fwd", 12)].max())
xs = gs(P, g, lb, ub, 3, order=list(range(6, -1, -1)), omega=1.5)          # noqa: F821   (reverse relaxed x 3)
time_of[("rsor", 3)] = pas(P, g, lb, ub, xs, (xs <= lb) | (xs >= ub), "worst")[0] * T_PASS + 3 * T_SWEEP   # noqa: F821
xs = gs(P, g, lb, ub, 3)                                                   # noqa: F821
time_of[("fwd", 3)] = pas(P, g, lb, ub, xs, (xs <= lb) | (xs >= ub), "worst")[0] * T_PASS + 3 * T_SWEEP    # noqa: F821
shipped = [("fwd", 3), ("rev", 6), ("sor", 12), ("rsor", 3)]
print("seed %d, %d instances: the shipped four (fwd x 3, rev x 6, relaxed x 12, reverse relaxed x 3): %.2f us" % (
    SEED, B, np.minimum.reduce([time_of[k] for k in shipped]).max()))
