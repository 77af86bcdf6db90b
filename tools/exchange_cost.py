#!/usr/bin/env python3
"""
What the step's collective costs, as far as ONE GPU can show it (profiles/r04_scaling_projection.txt):
  * RCCL all_gather_into_tensor at world size 1 (the floor of the call),
  * the direct exchange (bild_exchange_allgather) at world size 1 (a copy through the own receive block), and
  * the direct exchange between TWO PROCESSES that share the GPU (IPC-mapped receive blocks, flags, bounded polls: the
    protocol as it runs between GPUs, minus the xGMI hop),
each alone and behind the 10k x T=1000 likelihood step it follows in an AMIS iteration.
    python tools/exchange_cost.py            (spawns the second process itself)
"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np


def worker(base, rank, world, quiet):
    import torch
    torch.cuda.set_device(0)
    import helpers as H, bild_amd, bench
    from bild_amd import _lib, dist as bdist
    dev = torch.device('cuda', 0)
    n, T, k = 10000, 1000, 4
    model, trajs, ss, thetas = bench.build_workload(rank, n, T, k)
    h, ts = model.handle(), model.trajset(trajs[0])
    _lib.logl_st(h, ts, ss, thetas)
    d_ss = torch.from_numpy(np.ascontiguousarray(ss)).to(dev)
    d_th = torch.from_numpy(thetas.astype(np.uint8)).to(dev)
    d_out = torch.zeros(n, dtype=torch.float64, device=dev)
    d_all = torch.zeros(n * max(world, 1), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ex = bdist.DirectExchange.from_files(base, world, rank, n, nonce='cost') if world > 1 else bdist.DirectExchange(1, 0, n)

    def likelihood():
        _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), d_out.data_ptr(), stream=stream)

    def exchange():
        ex.allgather(d_out.data_ptr(), d_all.data_ptr(), n, stream)

    def timed(fns, steps=400, warm=20):
        for _ in range(warm):
            for f in fns:
                f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for f in fns:
                f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6
    out = {}
    # (every rank runs the same sequence: the exchanges pair up)
    out['exchange alone'] = timed([exchange])
    out['likelihood alone'] = timed([likelihood])
    out['likelihood + exchange'] = timed([likelihood, exchange])
    ex.status()
    if world == 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        rccl = lambda: dist.all_gather_into_tensor(d_all, d_out)
        out['rccl all_gather alone (world 1)'] = timed([rccl])
        out['likelihood + rccl all_gather (world 1)'] = timed([likelihood, rccl])
        dist.destroy_process_group()
    if not quiet:
        for key, us in out.items():
            print(f"  world {world}: {key:42s} {us:7.1f} us per step")


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == '1')
        sys.exit(0)
    print("shard: 10 000 doubles (80 KB) per rank; likelihood: 10 000 candidates x T = 1000, k = 4, rows resident in HBM")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([sys.executable, __file__, '--worker', os.path.join(tmp, 'w1'), '0', '1', '0'])
        base = os.path.join(tmp, 'w2')
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs = [subprocess.Popen([sys.executable, __file__, '--worker', base, str(r), '2', '1' if r else '0'], env=env) for r in range(2)]
        for p in procs:
            if p.wait(timeout=600) != 0:
                raise SystemExit("worker failed")
    print("(world 2: two processes share the GPU -- their likelihood kernels run side by side, so 'likelihood + exchange' there is an upper\n"
          " bound of contention, not a projection; 'exchange alone' is the protocol's cost without a wire: launch, 80 KB stored twice,\n"
          " release, flag, poll, acquire, 160 KB copied out)")
