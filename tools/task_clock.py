"""
Where does the launch's time go?  Needs a diagnostics build of the library (-DBILD_TASK_CLOCK: every task stamps the
100 MHz wall clock at its start and end); compares those stamps with the frames each task ran (from the normal build).

    python tools/ab.py build clock:-DBILD_TASK_CLOCK norm:
    python tools/task_clock.py [n] [T] [k]          # on the GPU box
"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, 'bild_amd', 'variants')

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ctypes
    import numpy as np, torch, helpers as H, bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    n, T, k, use_order = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rng = np.random.default_rng(2000)
    model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
    ss, th = H.candidate_profiles(rng, n, k, 2)
    a, b = segments_from_st(ss, th, T)
    h, ts = model.handle(), model.trajset(traj)
    dev = torch.device('cuda', 0)
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    frames = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr())   # builds the tables
    torch.cuda.synchronize()
    d_order = None
    order = np.arange(n)
    if use_order:
        o = _lib.schedule_segments(h, ts, a, b, None)
        if o is not None:
            order = np.asarray(o).astype(np.int64)
            d_order = torch.from_numpy(np.asarray(o)).to(dev)
    for _ in range(3):
        _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(),
                                  d_order=d_order.data_ptr() if d_order is not None else 0)
    torch.cuda.synchronize()
    _lib.lib().bild_debug_frames_per_task(ctypes.c_void_p(frames.data_ptr()))
    _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(),
                              d_order=d_order.data_ptr() if d_order is not None else 0)
    torch.cuda.synchronize()
    _lib.lib().bild_debug_frames_per_task(None)
    np.save(sys.argv[6], frames.cpu().numpy())
    np.save(sys.argv[6] + '.order.npy', order)
    sys.exit(0)

import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for use_order in (0, 1):
    res = {}
    names = ('norm', 'clock') + tuple(x for x in ('events', 'stages') if os.path.exists(os.path.join(VDIR, f'libbild_amd_{x}.so')))
    for name in names:
        path = f'/tmp/task_clock_{name}.npy'
        env = dict(os.environ, BILD_AMD_LIB=os.path.join(VDIR, f'libbild_amd_{name}.so'))
        r = subprocess.run([sys.executable, __file__, '--child', str(n), str(T), str(k), str(use_order), path], env=env,
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:]); sys.exit(1)
        res[name] = np.load(path)
    order = np.load('/tmp/task_clock_norm.npy.order.npy')
    f = res['norm'].astype(np.int64)
    c = res['clock'].astype(np.int64) & 0xffffffff
    b, e = (c >> 16) & 0xffff, c & 0xffff
    t0 = b[order[0]]                                    # the first slot's start is (about) the launch's
    b, e = ((b - t0) & 0xffff) / 100., ((e - t0) & 0xffff) / 100.   # us since then
    b[b > 500] -= 655.36; e[e > 500] -= 655.36
    dur = e - b
    print(f"n={n} T={T} k={k}  {'scheduler order' if use_order else 'array order'}: tasks start {b.min():.1f}..{b.max():.1f} us, "
          f"end {e.min():.1f}..{e.max():.1f} us  (mean end {e.mean():.1f}, p50 {np.median(e):.1f}, p90 {np.percentile(e, 90):.1f}, p99 {np.percentile(e, 99):.1f})")
    # by position: slot j of the launch -> wave j // 4
    fo, do, eo, bo = f[order], dur[order], e[order], b[order]
    nw = n // 4
    wf = fo[: nw * 4].reshape(nw, 4); wd = do[: nw * 4].reshape(nw, 4); we = eo[: nw * 4].reshape(nw, 4)
    wave_end = we.max(axis=1)
    print(f"   per wave: frames of its busiest row mean {wf.max(axis=1).mean():.1f}, sum over rows mean {wf.sum(axis=1).mean():.1f}; "
          f"wave time mean {wd.max(axis=1).mean():.1f} us, max {wd.max():.1f} us")
    # a linear model of a wave's time: a + b * (busiest row's frames) + c * (sum of the others')
    X = np.stack([np.ones(nw), wf.max(axis=1), wf.sum(axis=1) - wf.max(axis=1)], axis=1)
    coef, *_ = np.linalg.lstsq(X, wd.max(axis=1), rcond=None)
    print(f"   wave time ~ {coef[0]:.1f} us + {coef[1]:.3f} us x frames(busiest row) + {coef[2]:.3f} us x frames(other rows)")
    for lo, hi in ((0, 0), (1, 50), (51, 100), (101, 150), (151, 1000)):
        m = (wf.max(axis=1) >= lo) & (wf.max(axis=1) <= hi)
        if m.any():
            print(f"   waves whose busiest row ran {lo:3d}..{hi:4d} frames: {m.sum():5d}, time mean {wd.max(axis=1)[m].mean():6.1f} us, "
                  f"max {wd.max(axis=1)[m].max():6.1f} us, end mean {wave_end[m].mean():6.1f} us")
    last = np.argsort(-wave_end)[:8]
    for w in last:
        print(f"   late wave {w:5d}: rows' frames {wf[w].tolist()}, starts {bo[w * 4]:.1f} us, ends {wave_end[w]:.1f} us")
    if 'events' in res:     # -DBILD_TASK_CLOCK=2: ticks inside the comparison / jump blocks of the frame loop, and their number
        v = res['events'].astype(np.int64) & 0xffffffff
        ev_us, ev_n = ((v >> 16) & 0xffff) / 100., v & 0xffff
        busy = f > 0
        print(f"   comparisons+jumps per busy candidate: {ev_n[busy].mean():.1f}, {ev_us[busy].sum() / max(ev_n[busy].sum(), 1):.2f} us each; "
              f"share of the task's time {ev_us[busy].sum() / dur[busy].sum():.2f}")
        for lo, hi in ((1, 50), (51, 100), (101, 150), (151, 1000)):
            m = (f >= lo) & (f <= hi)
            if m.any():
                print(f"   candidates with {lo:3d}..{hi:4d} frames: {m.sum():5d}: task {dur[m].mean():6.1f} us, of it events {ev_us[m].mean():5.1f} us "
                      f"({ev_n[m].mean():.1f} of them), rest per frame {(dur[m].mean() - ev_us[m].mean() - 13.) / f[m].mean():.3f} us")
    if 'stages' in res:     # -DBILD_TASK_CLOCK=3: stamps inside the per-task prologue, 20 ns units since the task began
        v = res['stages'].astype(np.int64) & 0x3fffffff
        ta, tb, tc = ((v >> 20) & 0x3ff) / 50., ((v >> 10) & 0x3ff) / 50., (v & 0x3ff) / 50.
        idle = f == 0
        print(f"   prologue of candidates that run no frame (us since the task began): descriptor + segment list {ta[idle].mean():.2f}, "
              f"state vectors {tb[idle].mean():.2f}, tables walked {tc[idle].mean():.2f}, result written {dur[idle].mean():.2f}")
