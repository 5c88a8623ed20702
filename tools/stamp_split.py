"""Diagnostic: per-wave cycle stamps of the opt-in role-split kernel (CLIK_ROLE_SPLIT=1):
main wave of mode 0 (wave 0) and its helper (wave 1).  Separate JIT build with stamps."""
import ctypes as C, os, sys
os.environ["CLIK_NO_AOT"]="1"; os.environ["CLIK_JIT_STAMPS"]="1"; os.environ["CLIK_ROLE_SPLIT"]="1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, torch
import casclik_amd as cc
from casclik_amd import skills, jit
fk=skills.iiwa()
ctrl=cc.PseudoInverseController(skill_spec=skills.stack_skill(fk), options=dict(skills.STACK_OPTIONS)); ctrl.setup_problem_functions()
B=16384
for dist in ("interior","mixed"):
    Q,Y=skills.synthetic_inputs(fk,B,seed=0,distribution=dist)
    Qd,Yd=torch.from_numpy(Q).cuda(),torch.from_numpy(Y).cuda()
    tick=ctrl.bind_batch(Qd,input_var=Yd)
    for _ in range(100): tick()
    torch.cuda.synchronize()
    lib=jit.attach.last_library
    n=8*(B//64); buf=(C.c_ulonglong*n)()
    lib.clik_jit_read_stamps.argtypes=[C.POINTER(C.c_ulonglong),C.c_int]
    assert lib.clik_jit_read_stamps(buf,n)==0
    st=np.array(buf[:],dtype=np.float64).reshape(-1,8)
    m=lambda a: np.median(a)
    print(dist,"wave0: prologue %.0f FK %.0f mode(main0) %.0f tail %.0f total %.0f | helper: FK done at %.0f, factor published at %.0f (rel. to wave0 start)"%(
        m(st[:,1]-st[:,0]), m(st[:,2]-st[:,1]), m(st[:,3]-st[:,2]), m(st[:,5]-st[:,3]), m(st[:,5]-st[:,0]), m(st[:,6]-st[:,0]), m(st[:,7]-st[:,0])))
