""" device-side draws (FixedkSampler(rng='device')): moments for a non-uniform proposal, and evidence spread vs the NumPy stream """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib

rng = np.random.default_rng(11)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 250, 2, 60), rng=rng)
a0 = np.array([0.3, 2.5, 7.0, 40.0])
logp0 = np.log(np.array([[0.2, 0.5, 0.5, 0.5], [0.8, 0.5, 0.5, 0.5]]))
core = _lib.AmisCore(model.transitions, a0, logp0, 1e-2, 1e-3, 0.0)
core.use_device(True)
N = 200000
core.step_device_rng(model.handle(), model.trajset(traj), N, 12345)
ss, th = core.pool_samples()
A = a0.sum()
print("mean  ", ss.mean(axis=0), "exact", a0 / A)
print("var   ", ss.var(axis=0), "exact", a0 * (A - a0) / (A * A * (A + 1)))
print("first state 1 share", th[:, 0].mean(), "exact 0.8")
ref = np.random.default_rng(1).dirichlet(a0, size=N)
for j in range(4):
    q = [0.01, 0.1, 0.5, 0.9, 0.99]
    print(f"  quantiles coord {j}: device {np.quantile(ss[:, j], q)}  numpy {np.quantile(ref[:, j], q)}")

k, Ns = 3, 4000
def run(seed, which, steps=6):
    np.random.seed(seed)
    s = bild_amd.FixedkSampler(traj, model, k=k, N=Ns, max_fev=10 ** 9, max_fcomplete=0, rng=which, seed=seed)
    for _ in range(steps):
        s.step()
    return s
for steps in (2, 6, 12):
    ev = {w: np.array([run(100 + sd + (1000 if w == 'numpy2' else 0), 'numpy' if w.startswith('numpy') else w, steps).evidences[-1][:2] for sd in range(12)])
          for w in ('numpy', 'numpy2', 'device')}
    for w in ev:
        print(f"steps={steps} {w:7s}: logev mean {ev[w][:, 0].mean():.3f} sd over seeds {ev[w][:, 0].std():.3f}  mean reported dlogev {ev[w][:, 1].mean():.3f}")
