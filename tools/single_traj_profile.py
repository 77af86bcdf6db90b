#!/usr/bin/env python3
""" Where `sample(traj, model)` through the native inference driver spends its time, one trajectory after the other (fresh trajectories):
    the three NumPy draws, the rounds (likelihood call + bookkeeping), the plans.    python tools/single_traj_profile.py [seed] [n_traj + 1] """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import core, _lib
rng = np.random.default_rng(6)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, int(rng.integers(150, 601)), 2, 120), rng=rng) for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12)]
bild_amd.sample(trajs[0], model)
acc = {'gamma':0.0,'uniform':0.0,'normal':0.0,'round':0.0,'plan':0.0}
cnt = {'normals':0,'gammas':0,'rounds':0}
og, ou, on = np.random.standard_gamma, np.random.random_sample, np.random.standard_normal
def tg(*a):
    t=time.perf_counter(); r=og(*a); acc['gamma']+=time.perf_counter()-t; cnt['gammas']+=np.size(r); return r
def tu(*a):
    t=time.perf_counter(); r=ou(*a); acc['uniform']+=time.perf_counter()-t; return r
def tn(*a):
    t=time.perf_counter(); r=on(*a); acc['normal']+=time.perf_counter()-t; cnt['normals']+=np.size(r); return r
np.random.standard_gamma, np.random.random_sample, np.random.standard_normal = tg, tu, tn
orig_round, orig_plan = _lib.RunHandle.round, _lib.RunHandle.plan
def tr(self,*a,**k):
    t=time.perf_counter(); r=orig_round(self,*a,**k); acc['round']+=time.perf_counter()-t; cnt['rounds']+=1; return r
def tp(self,*a,**k):
    t=time.perf_counter(); r=orig_plan(self,*a,**k); acc['plan']+=time.perf_counter()-t; return r
_lib.RunHandle.round, _lib.RunHandle.plan = tr, tp
np.random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
t0=time.perf_counter()
res=[bild_amd.sample(t, model) for t in trajs[1:]]
wall=time.perf_counter()-t0
steps=sum(len(r.log['k']) for r in res)
print(f"{len(res)} trajectories, {steps} steps, {cnt['rounds']} rounds: wall {wall*1e3:.1f} ms = {wall/steps*1e6:.0f} us per step")
for k,v in acc.items(): print(f"  {k:8s} {v*1e3:8.2f} ms  ({v/wall*100:4.1f} %)")
print(f"  normals drawn {cnt['normals']}, gammas {cnt['gammas']}; other {(wall-sum(acc.values()))*1e3:.1f} ms")
