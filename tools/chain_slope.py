"""
Cost of a frame inside a chain of close switches, in situ: 64 identical candidates (each alone on a wave of the frame loop over
the work lists), K switches g frames apart; device time of the frame-loop kernel against K gives microseconds per frame
(basis change every g frames included) and the fixed part of a task.     python tools/chain_slope.py [g]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib

g = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2   # states of the model; the candidates cycle through them
T, n = 1000, 64
rng = np.random.default_rng(2000)
model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1, **(dict(looppositions=H.LOOPS[S]) if S != 2 else {}))
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, T // 5), rng=rng)
h, ts = model.handle(), model.trajset(traj)
dev = torch.device('cuda', 0)
rows = []
for K in (3, 5, 7, 9, 11, 13, 15):
    a = np.zeros((n, K + 1), dtype=np.int32)
    a[:, 1:] = 100 + g * np.arange(K)[None, :]
    b = np.zeros((n, K + 1), dtype=np.int32)
    b[:, :] = (np.arange(K + 1) % S)[None, :]
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    def go():
        _lib.logl_segments_device(h, ts, n, K + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(True)
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(False)
    ms, c, _ = _lib.kernel_timing_read()
    wms, wc = _lib.kernel_timing_read_walk()
    fr = _lib.frames_run_read(h) / (20.0 * n)
    rows.append((K, fr, ms / c * 1e3))
    print(f"K={K:2d} switches {g} frames apart: frames run per candidate {fr:6.1f}, frame-loop kernel {ms / c * 1e3:7.2f} us, walk {wms / max(wc, 1) * 1e3:5.2f} us", flush=True)
x = np.array([r[1] for r in rows]); y = np.array([r[2] for r in rows])
slope, icpt = np.polyfit(x, y, 1)
print(f"fit: {icpt:.2f} us + {slope:.4f} us per frame (one basis change per {g} frames included)")
