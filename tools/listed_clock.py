"""
The listed frame loop, task by task (diagnostics builds with -DBILD_TASK_CLOCK=1 / =3 in bild_amd/variants/): when every
listed task of the headline batch begins and ends (100 MHz wall clock), how many frames it ran, and the stamps inside its
prologue.      python tools/listed_clock.py [n] [T] [k]
"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, 'bild_amd', 'variants')
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ctypes
    import numpy as np, torch, helpers as H, bild_amd, bench
    from bild_amd import _lib
    n, T, k = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    model, trajs, ss, th = bench.build_workload(0, n, T, k)
    h, ts = model.handle(), model.trajset(trajs[0])
    dev = torch.device('cuda', 0)
    _lib.logl_st(h, ts, ss[:100], th[:100])
    d_ss, d_th = torch.from_numpy(np.ascontiguousarray(ss)).to(dev), torch.from_numpy(th.astype(np.uint8)).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    frames = torch.full((n,), -1, dtype=torch.int32, device=dev)
    go = lambda: _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    _lib.lib().bild_debug_frames_per_task(ctypes.c_void_p(frames.data_ptr()))
    go()
    torch.cuda.synchronize()
    _lib.lib().bild_debug_frames_per_task(None)
    np.save(sys.argv[5], frames.cpu().numpy())
    sys.exit(0)
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
res = {}
for name, lib in (('norm', os.path.join(ROOT, 'bild_amd', 'libbild_amd.so')), ('clock', os.path.join(VDIR, 'libbild_amd_clock.so')),
                  ('stages', os.path.join(VDIR, 'libbild_amd_stages.so'))):
    path = f'/tmp/listed_clock_{name}.npy'
    r = subprocess.run([sys.executable, __file__, '--child', str(n), str(T), str(k), path], env=dict(os.environ, BILD_AMD_LIB=lib), capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    res[name] = np.load(path)
f = res['norm'].astype(np.int64)
listed = f > 0
c = res['clock'].astype(np.int64) & 0xffffffff
b, e = (c >> 16) & 0xffff, c & 0xffff
t0 = b[listed].min()
b, e = ((b - t0) & 0xffff) / 100., ((e - t0) & 0xffff) / 100.
dur = e - b
v = res['stages'].astype(np.int64) & 0x3fffffff
ta, tb, tc = ((v >> 20) & 0x3ff) / 50., ((v >> 10) & 0x3ff) / 50., (v & 0x3ff) / 50.
print(f"n={n} T={T} k={k}: {listed.sum()} listed tasks, frames run mean {f[listed].mean():.1f} max {f[listed].max()}; tasks begin {b[listed].min():.1f}..{b[listed].max():.1f} us, "
      f"end {e[listed].min():.1f}..{e[listed].max():.1f} us (p50 {np.median(e[listed]):.1f}, p90 {np.percentile(e[listed], 90):.1f})")
print(f"  prologue stamps of listed tasks (us since the task began): list cleaned {ta[listed].mean():.2f}, state vectors + plan {tb[listed].mean():.2f}, tables walked {tc[listed].mean():.2f}")
X = np.stack([np.ones(listed.sum()), f[listed]], axis=1)
coef, *_ = np.linalg.lstsq(X, dur[listed], rcond=None)
print(f"  task time ~ {coef[0]:.1f} us + {coef[1]:.3f} us x frames")
order = np.argsort(-e * listed)[:10]
for i in order:
    print(f"  late task {i:5d}: {f[i]:4d} frames, begins {b[i]:5.1f}, ends {e[i]:5.1f} us ({dur[i]:.1f} us; {(dur[i] - coef[0]) / max(f[i], 1):.3f} us per frame beyond the fit's offset); prologue {tc[i]:.1f} us")
