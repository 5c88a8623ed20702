"""Tick time of the UR5 distance skill with 1..5 joint-limit SetConstraints (2..32 modes):
run-time instantiated static kernels vs the dynamic kernel (CLIK_FORCE_DYNAMIC=1).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, torch
import casclik_amd as cc
from casclik_amd import skills, sym as cs
fk=skills.ur5()
t = cs.MX.sym("t"); q = cs.MX.sym("q", 6)
p = fk["T_fk"](q)[:3, 3]
lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
def bench(nsets,B=16384):
    cons = [cc.EqualityConstraint("dist", cs.norm_2(np.array([0.5, 0.5, 0.5]) - p), gain=50.0, constraint_type="soft", priority=6)]
    for i in range(nsets):
        cons.append(cc.SetConstraint("limit_q_%d" % i, q[i], set_min=0.3 * lo[i], set_max=0.3 * hi[i], priority=i))
    spec = cc.SkillSpecification("point", t, q, constraints=cons)
    ctrl = cc.PseudoInverseController(skill_spec=spec); ctrl.setup_problem_functions()
    rng=np.random.default_rng(4)
    Q = rng.uniform(0.35 * lo, 0.35 * hi, size=(B, 6))
    Qd=torch.from_numpy(Q).cuda(); dQ=torch.empty_like(Qd)
    tick=ctrl.bind_batch(Qd,out=dQ)
    for _ in range(50): tick()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(300): tick()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/300
    print("sets %d modes %d kernel %s: %.1f us/tick"%(nsets, ctrl.n_modes, ctrl.kernel_name, dt*1e6))
for n in (1,2,3,4,5): bench(n)
