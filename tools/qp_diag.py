import os, sys
os.environ["CLIK_NO_AOT"]="1"; os.environ["CLIK_JIT_DEFINES"]="-DCLIK_QP_DIAG"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, torch
import casclik_amd as cc
from casclik_amd import skills
fk=skills.iiwa()
ctrl=cc.ReactiveQPController(skill_spec=skills.qp_skill(fk)); ctrl.setup_problem_functions(); ctrl.setup_solver()
print(ctrl.kernel_name)
for dist in ("mixed","interior"):
    Q,Y=skills.synthetic_inputs(fk,16384,seed=0,distribution=dist)
    dq,_,sl,st=ctrl.solve_batch(0.0,Q,input_var=Y)
    it=(st>>8)&255; wp=(st>>16)&255; cold=(st>>24)&7
    print(dist,"fell back to the cold start: unsound %d, emptied %d, too large %d; iterations of those lanes %s" % (
        (cold&1).astype(bool).sum(), (cold&2).astype(bool).sum(), (cold&4).astype(bool).sum(), np.bincount(it[cold!=0])))
    print(dist,"status",np.bincount(st&255),"iters per lane",np.bincount(it),"warm passes",np.bincount(wp))
    print("  per-wave max iters",np.bincount(it.reshape(-1,64).max(axis=1)),"per-wave max warm",np.bincount(wp.reshape(-1,64).max(axis=1)))
