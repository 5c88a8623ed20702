#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/.

The reference (casclik on casadi==3.4.1 + urdf2casadi) cannot be imported in the
build container, so these vectors are ORACLE-OF-RECORD: produced by the numpy
restatement oracle/clik_oracle.py, NOT by CasADi (parity unpinned, see the
oracle header).  They freeze the oracle's answers so that (a) oracle edits are
visible as fixture diffs and (b) the GPU tests have inputs/outputs that do not
depend on the oracle being importable.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from casclik_amd import skills                 # noqa: E402
from oracle import clik_oracle                 # noqa: E402

CASES = {
    "iiwa_position": ("iiwa", skills.position_skill, None, 3),
    "iiwa_pose": ("iiwa", skills.pose_skill, None, 7),
    "iiwa_stack": ("iiwa", skills.stack_skill, skills.STACK_OPTIONS, 7),
    "ur5_stack": ("ur5", skills.stack_skill, skills.STACK_OPTIONS, 7),
}


def main():
    fks = {"iiwa": skills.iiwa(), "ur5": skills.ur5()}
    out = {}
    for name, (robot, make, opts, ny) in CASES.items():
        fk = fks[robot]
        spec = make(fk)
        Qi, Yi = skills.synthetic_inputs(fk, 12, seed=7, distribution="interior")
        Qm, Ym = skills.synthetic_inputs(fk, 20, seed=8, distribution="mixed")
        Q = np.vstack([Qi, Qm])
        Y = np.vstack([Yi, Ym])[:, :ny]
        dq, mode = clik_oracle.pinv_solve_batch(spec, opts, 0.0, Q, Y=Y)
        out[name + "_Q"] = Q
        out[name + "_Y"] = Y
        out[name + "_dq"] = dq
        out[name + "_mode"] = mode
    fk = fks["iiwa"]
    spec = skills.qp_skill(fk)
    Q, Y = skills.synthetic_inputs(fk, 24, seed=9, distribution="interior")
    dq, _, slack, status = clik_oracle.qp_solve_batch(spec, 0.0, Q, Y=Y)
    assert (status == 0).all()
    out["iiwa_qp_Q"], out["iiwa_qp_Y"] = Q, Y
    out["iiwa_qp_dq"], out["iiwa_qp_slack"] = dq, slack
    np.savez_compressed(os.path.join(HERE, "clik_golden.npz"), **out)
    print("wrote clik_golden.npz with", sorted(out))


if __name__ == "__main__":
    main()
