#!/usr/bin/env python3
"""
SURVEY 8d "CPU baseline beside it", all host cores: one process per core, each running the CPU kernel in the
reference's Python loop (amis.py:735-739) over its own shard of the bench batch (10 000 x T = 1000, 2-state N = 20)
for a fixed time.  Kernels: the reference's Cython MSRouse_logL built unmodified into oracle/_ref ("reference"),
and this repository's C restatement of it ("port", oracle/msrouse_logl.c).  CPU only -- nothing here touches a GPU.

    python tests/tools/cpu_allcores.py [processes] [seconds]
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, kind, seconds, out):
    for var in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[var] = '1'
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    import helpers as H
    import bench
    from oracle import oracle
    T = 1000
    model, traj, ss, thetas = bench.build_workload(0, 10000, T, 4)
    lo, hi = rank * (10000 // world), (rank + 1) * (10000 // world)
    states = H.expand(ss[lo:hi], thetas[lo:hi], T)
    if kind == 'reference':
        ref = oracle.load_reference_cython()
        if ref is None:
            out.put((rank, kind, None, 0.0))
            return

        class M:
            pass
        m = M()
        m.models, m.measurement, m.d, m._get_noise = model.models, model.measurement, model.d, model._get_noise

        def one(i):
            return ref(m, H.ProfileView(states[i]), traj)
    else:
        arrays, w, err, x = model.arrays(), model.measurement, model.localization_error, traj[:]

        def one(i):
            return oracle.logl(arrays, w, err, x, states[i])
    one(0)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        one(n % len(states))
        n += 1
    out.put((rank, kind, n, time.perf_counter() - t0))


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    ctx = mp.get_context('spawn')
    for kind in ('reference', 'port'):
        for world in sorted({1, procs}):
            q = ctx.Queue()
            ps = [ctx.Process(target=worker, args=(r, world, kind, seconds, q)) for r in range(world)]
            for p in ps:
                p.start()
            res = [q.get() for _ in ps]
            for p in ps:
                p.join()
            if any(r[2] is None for r in res):
                print(f"{kind}: oracle/_ref not built here, skipped")
                break
            rate = sum(r[2] / r[3] for r in res)
            print(f"{kind:9s} x {world:3d} process(es): {rate:9.1f} evals/s  ({rate / world:7.1f} per process), "
                  f"{seconds:.0f} s each, T=1000, N=20, d=3", flush=True)


if __name__ == '__main__':
    main()
