#!/usr/bin/env python3
"""What the Moe-2016 figure pins (tests/golden/moe_figure_pins.py) resolve: deliberately wrong PseudoInverseControllers
(oracle/clik_oracle.py `_wrong`, changed gains / options) through the notebook's loop against the same pins.
    python tools/moe_sensitivity.py [ticks = 2500]          (10000 = the whole stored run, ~20 s per line)"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import notebook_figures as cf
from oracle import clik_oracle
fk=cf.moe_fk(); pos=lambda q: fk["chain"].fk_numeric(q)[:3,3]
def run(case, n_ticks, wrong=None, mutate=None, options=None):
    kind,sit=case.split('_'); spec=cf.moe_skill(fk,sit)
    if mutate: mutate(spec)
    opts=dict(cf.moe_options(case) or {}); opts.update(options or {})
    def solve(t,q):
        dz,mode=clik_oracle.pinv_solve_batch(spec, opts, float(t), q[None,:], _wrong=wrong)
        return dz[0], int(mode[0])
    return cf.simulate_moe(solve,pos,n_ticks)
def gain(label,v):
    def m(spec):
        for c in spec.constraints:
            if c.label==label: c.gain=v
    return m
def report(name, case, res):
    t,q,p,e,m=res
    pins=cf.moe_pins(case,t,p,e,m)
    pins=[x for x in pins if x[2]>0]
    worst=max(pins,key=lambda x:x[1])
    print('%-34s %-14s worst %.2f px (%s at t=%s); >1px: %s'%(name,case,worst[1],worst[0],worst[3],[(k.replace('moe_',''),round(w,1)) for k,w,n,tt in pins if w>1.0]))
N=int(sys.argv[1]) if len(sys.argv)>1 else 2500
for case in ('pinv_multidim','pinv_singular'):
    report('literal',case,run(case,N))
    for w in ('no_S','no_D1','textbook_projection','active_first'):
        if w=='no_S' and case!='pinv_multidim': continue
        report(w,case,run(case,N,wrong=w))
    report('gain 0.18',case,run(case,N,mutate=gain('move_point2',0.18)))
    report('feedforward off',case,run(case,N,options={'feedforward':False}))
    report('damping 1e-3',case,run(case,N,options={'damping_factor':1e-3}))
