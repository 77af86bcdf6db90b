#!/usr/bin/env python3
"""
BASELINE configs[4] with trajectories sharded over ranks (`dist.sample_many_distributed`): one process per GPU on
a multi-GPU node; on a one-GPU box the ranks can share the card to rehearse (gloo), which still parallelises the
host-side bookkeeping.

    python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 tools/config5_dist.py [n_traj] [backend]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.distributed as dist
import helpers as H, bild_amd
from bild_amd.dist import sample_many_distributed

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 500
backend = sys.argv[2] if len(sys.argv) > 2 else 'gloo'
rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))
dist.init_process_group(backend)
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = []
for j in range(n_traj):
    T = int(rng.integers(150, 601))
    trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 120), rng=rng))
kw = dict(init_runs=5, k_max=6, sampler_kw={'N': 100, 'max_fev': 3000}, choice_kw={'samplesize': 2000})
model.logL_segments(np.zeros((1, 1), np.int32), np.zeros((1, 1), np.int32), trajs[:1], np.zeros(1, np.int32))  # library + device up
dist.barrier()
t0 = time.perf_counter()
res = sample_many_distributed(trajs, model, seed=11, gather='root', **kw)
t1 = time.perf_counter()
dist.barrier()
t2 = time.perf_counter()
if rank == 0:
    ks = [int(r.best_k()) for r in res]
    steps = sum(len(s.samples) for r in res for s in r.samplers)
    print(f"{world} rank(s), {n_traj} trajectories: {t2 - t0:.2f} s wall incl. gathering the results on rank 0 "
          f"(rank 0 alone {t1 - t0:.2f} s), {steps} AMIS steps, best k histogram {np.bincount(ks).tolist()}")
dist.destroy_process_group()
