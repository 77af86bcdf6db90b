#!/usr/bin/env python3
"""
Headline benchmark: logL evaluations / second of one AMIS batch on MI355X.

Workload (BASELINE.json configs[1]): a batch of 10 000 candidate looping profiles x 1 trajectory, T = 1000 frames,
2-state Rouse model, N = 20 monomers, d = 3, one localization error (d* = 1), k = 4 switches per candidate, fp64.
One "step" = one pass of the hot path over one such batch: everything `FixedkSampler.logL(ss, thetas)` does for one
AMIS iteration (reference bild/amis.py:717-739).

What the JSON line reports
  value / ms_per_step   the batch with candidates and trajectory resident in HBM (`bild_logl_segments_device`,
                        nothing crosses PCIe inside the timed region) -- the bench contract's definition;
  api_seam              the SAME batch through the seam the path is defined by: `FixedkSampler.logL(ss, thetas)` with
                        host arrays in and a host array out per step (native (s, theta) -> segment conversion, one
                        packed H2D copy out of pinned memory, the launch, one D2H copy, one synchronisation);
  roofline              the dominant kernel against the fp64 vector peak: `achieved` / `frac` count the operations the
                        kernel really executes (frames skipped through shared prefixes are not counted);
                        `canonical_equiv_frac` prices the same time against the reference's dense operation count
                        (a speed-up figure, may exceed 1, never a utilisation);
  amis_step             a whole AMIS iteration around the seam (draws, likelihood, refit over all samples drawn so far);
  large_batch           twenty times the batch on the same trajectory: the throughput regime of the same kernel;
  cpu_baseline          the reference's own Cython kernel on one host core (and on all cores, secondary).

Multi-GPU: one process per GPU.  `--scaling weak` (default, what the driver runs): every rank evaluates its own 10k
batch and joins ONE all-gather of the log-likelihoods per step (RCCL), as an AMIS step needs them to form the
importance weights.  `--scaling strong`: a FIXED workload is partitioned over the ranks -- `--strong-config 1`:
configs[1] (10k x 1 trajectory, contiguous sample shards), `--strong-config 2`: configs[2] (256 trajectories x 1000
samples, whole trajectories per rank, `dist.shard_by_trajectory`) -- again one all-gather per step.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (= fp64 matrix) peak, AMD spec
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def build_workload(seed, n_samples, T, k, S=2, N=20, d=3, err=0.1, n_traj=1):
    """ model, n_traj trajectories, and n_samples candidates per trajectory """
    import helpers as H
    import bild_amd
    rng_t = np.random.default_rng(1000 + seed)
    rng_c = np.random.default_rng(2000 + seed)
    model = bild_amd.MultiStateRouse(N, 1., 5., d=d, looppositions=H.LOOPS[S], localization_error=err)
    trajs = []
    for _ in range(n_traj):
        truth = H.random_profile(rng_t, T, S, T // 5)
        trajs.append(model.trajectory_from_loopingprofile(truth, rng=rng_t))
    ss, thetas = H.candidate_profiles(rng_c, n_samples * n_traj, k, S)
    return model, trajs, ss, thetas


def _cpu_baseline_loop(model, traj, ss, thetas, T, budget_s, first):
    """ the timed loop itself; runs in a child process (see cpu_baseline) """
    import helpers as H
    from oracle import oracle
    ref = oracle.load_reference_cython()
    kind = 'reference'
    states = H.expand(ss[first:first + 8192], thetas[first:first + 8192], T)

    class M:
        pass
    m = M()
    m.models, m.measurement, m.d = model.models, model.measurement, model.d
    m._get_noise = model._get_noise
    if ref is None:
        kind = 'port'

        def ref(mm, prof, tr):
            return oracle.logl(model.arrays(), model.measurement, model.localization_error, tr[:], prof[:])
    ref(m, H.ProfileView(states[0]), traj)  # warm
    out = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and len(out) < len(states):
        out.append(ref(m, H.ProfileView(states[len(out)]), traj))
    return kind, time.perf_counter() - t0, np.array(out)


def _cpu_baseline_child(argv):
    """ `bench.py --cpu-baseline-child seed n T k S budget out.npz first`: CPU only, never touches the GPU """
    seed, n, T, k, S = (int(v) for v in argv[:5])
    model, trajs, ss, thetas = build_workload(seed, n, T, k, S=S)
    kind, dt, out = _cpu_baseline_loop(model, trajs[0], ss, thetas, T, float(argv[5]), int(argv[7]))
    np.savez(argv[6], kind=kind, dt=dt, out=out)


def cpu_baseline(seed, n, T, k, S, budget_s=15.0, procs=1):
    """
    The reference's own Cython kernel (compiled unmodified into oracle/_ref) on host cores, driven exactly like
    FixedkSampler.logL drives it (a Python loop, amis.py:735-739), on a bounded sample of the same batch.  Every
    process is a fresh child with the BLAS pools pinned to one thread from the start: inside this process (torch
    loaded, pools limited after the fact) the same loop is 20-35 % slower, which would flatter the GPU.
    `procs` > 1: that many children at once over disjoint parts of the batch (one per core).
    """
    import subprocess
    import tempfile
    env = dict(os.environ, OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
    with tempfile.TemporaryDirectory() as tmp:
        children = []
        for c in range(procs):
            path = os.path.join(tmp, f'cpu_baseline_{c}.npz')
            first = (c * (n // procs)) if procs > 1 else 0
            children.append((path, subprocess.Popen(
                [sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', str(seed), str(n), str(T), str(k),
                 str(S), str(budget_s), path, str(first)], env=env)))
        total, dts, outs, kind = 0, [], [], 'port'
        for path, child in children:
            if child.wait(timeout=budget_s + 600) != 0:
                raise RuntimeError("cpu baseline child failed")
            with np.load(path) as z:
                kind = str(z['kind'])
                dts.append(float(z['dt']))
                outs.append(np.array(z['out']))
                total += len(z['out'])
    what = 'reference Cython MSRouse_logL (oracle/_ref)' if kind == 'reference' else 'oracle C port'
    return dict(value=total / max(dts), unit='evals/s', cores=procs, kind=kind,
                sample=f"{total} profiles of the rank-0 batch, T={T}, {max(dts):.1f} s of {what} in a Python loop "
                       f"(amis.py:735-739), {procs} process(es), BLAS pinned to 1 thread each"), outs[0]


def _cpu_share():
    """ host cores this process may really use: the affinity mask, capped by the cgroup CPU quota where there is one """
    n = len(os.sched_getaffinity(0))
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--samples', type=int, default=10000, help='candidate profiles per GPU per step (weak scaling)')
    ap.add_argument('--T', type=int, default=1000)
    ap.add_argument('--k', type=int, default=4)
    ap.add_argument('--states', type=int, default=2)
    ap.add_argument('--path', default='auto', choices=['auto', 'modal', 'dense'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--strong-config', type=int, default=2, choices=[1, 2],
                    help='strong scaling: 1 = configs[1] (10k x 1 trajectory) split over the ranks, '
                         '2 = configs[2] (256 trajectories x 1000 samples) sharded by trajectory')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-allcores', action='store_true', help='also time the reference kernel on every host core')
    ap.add_argument('--no-reduce', action='store_true', help='keep all N modes (skip the invariant-subspace reduction)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the dense / canonical-path side measurements')
    ap.add_argument('--no-seam', action='store_true', help='skip the api_seam measurement (profiling runs that want the headline launches alone)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="collective backend; 'gloo' (log-likelihoods staged through host memory) rehearses the "
                         "multi-rank path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    n_dev = torch.cuda.device_count()
    if args.backend == 'nccl' and world > n_dev:
        raise SystemExit(f"{world} ranks but {n_dev} GPUs (use --backend gloo to rehearse)")
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group('gloo')
    # the native library is built in-tree by rank 0 only (normally a no-op: the .so travels with the repo)
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()

    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    from bild_amd import dist as bdist

    T, k = args.T, args.k
    dev = torch.device('cuda', dev_index)

    # ---- the workload of this rank ------------------------------------------------------------------------
    if args.scaling == 'weak':
        n = args.samples
        model, trajs, ss, thetas = build_workload(rank, n, T, k, S=args.states)
        traj_id = None
        n_global = n * world
        sizes = [n] * world
        workload = (f'configs[1]: {n} profile samples x 1 trajectory per GPU, T={T}, {args.states}-state Rouse N=20 d=3 '
                    f'd*=1, k={k} switches, fp64')
    elif args.strong_config == 1:
        n_global = args.samples
        model, trajs, ss, thetas = build_workload(0, n_global, T, k, S=args.states)
        lo, hi = bdist.shard_bounds(n_global, world, rank)
        ss, thetas, traj_id = ss[lo:hi], thetas[lo:hi], None
        n = hi - lo
        sizes = [b - a for a, b in (bdist.shard_bounds(n_global, world, r) for r in range(world))]
        workload = (f'configs[1] strong: {n_global} profile samples x 1 trajectory split over {world} GPU(s), T={T}, '
                    f'{args.states}-state, k={k}, fp64')
    else:
        n_traj_total, per = 256, 1000
        model, trajs_all, ss_all, thetas_all = build_workload(0, per, T, k, S=args.states, n_traj=n_traj_total)
        owners = bdist.shard_by_trajectory([T] * n_traj_total, [per] * n_traj_total, world)
        mine = owners[rank]
        trajs = [trajs_all[j] for j in mine]
        rows = np.concatenate([np.arange(j * per, (j + 1) * per) for j in mine])
        ss, thetas = ss_all[rows], thetas_all[rows]
        traj_id = np.repeat(np.arange(len(mine)), per).astype(np.int32)
        n = len(rows)
        n_global = n_traj_total * per
        sizes = [len(o) * per for o in owners]
        workload = (f'configs[2] strong: {n_traj_total} trajectories x {per} samples sharded by trajectory over {world} '
                    f'GPU(s) ({len(mine)} trajectories on this rank), T={T}, {args.states}-state, k={k}, fp64')

    model.path = args.path
    if args.no_reduce:
        a_ = model.arrays()
        model._handle = _lib.ModelHandle(a_['B'], a_['G'], a_['Sig'], a_['M0'], a_['C0'], model.measurement, reduce=False)
    h = model.handle()
    ts = model.trajset(trajs if traj_id is not None else trajs[0])      # trajectories resident in HBM
    seg_start, seg_state = segments_from_st(ss, thetas, T)
    d_start = torch.from_numpy(seg_start).to(dev)                        # candidates resident in HBM
    d_state = torch.from_numpy(seg_state).to(dev)
    d_tid = torch.from_numpy(traj_id).to(dev) if traj_id is not None else None
    # launch order of the resident candidates (a derived descriptor like the segments themselves: computed on the host
    # from the same (s, theta) batch, resident in HBM before the timed region; the api_seam figure includes computing it)
    _lib.logl_segments(h, ts, seg_start, seg_state, traj_id, path=args.path)   # one evaluation of the batch: the set's tables exist now
    order = _lib.schedule_segments(h, ts, seg_start, seg_state, traj_id, path=args.path)
    d_order = None if np.array_equal(order, np.arange(len(order))) else torch.from_numpy(order).to(dev)   # identity: nothing to pass
    pad = max(sizes)
    d_out = torch.zeros(pad, dtype=torch.float64, device=dev)
    d_all = torch.empty(pad * world, dtype=torch.float64, device=dev)

    def step(path, prefix=True):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.logl_segments_device(h, ts, n, k + 1, d_start.data_ptr(), d_state.data_ptr(),
                                  d_tid.data_ptr() if d_tid is not None else 0, d_out.data_ptr(), stream=stream, path=path,
                                  d_order=d_order.data_ptr() if (prefix and d_order is not None) else 0, prefix=prefix)
        if world > 1:                                 # the one collective of an AMIS step
            if args.backend == 'nccl':
                bdist.all_gather_logl(d_out, d_all)
            else:
                d_all.copy_(bdist.all_gather_logl(d_out.cpu()))

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        _lib.kernel_timing(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        _lib.kernel_timing(False)
        kms, launches, kname = _lib.kernel_timing_read()
        timed.frames_per_launch = _lib.frames_run_read(h) / max(launches, 1)   # counted on the device by the tasks themselves
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms / max(launches, 1), kname

    dt, kernel_ms, kname = timed(lambda: step(args.path), args.steps, args.warmup)
    value = n_global * args.steps / dt
    frames_total = float(sum(len(trajs[j]) for j in (traj_id if traj_id is not None else np.zeros(n, dtype=int))))
    frames_frac = timed.frames_per_launch / frames_total

    # ---- roofline of the dominant kernel: executed operations against the fp64 vector peak --------------------
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r02_hbm_traffic.json')) as f:
            tj = json.load(f)
        if tj['workload'] == {'samples': n, 'T': T, 'k': k, 'path': args.path} and args.states == 2:
            traffic = tj['traffic_bytes_corrected']
    except Exception:
        pass
    can, exe = _lib.flop_count(h, ts, n, traj_id=traj_id, path=args.path)
    exe *= frames_frac
    prefix_bytes, prefix_ms = _lib.prefix_info(ts)
    ksec = kernel_ms * 1e-3
    is_mfma = 'mfma' in kname
    alg_bytes = n * (k + 1) * 8 + n * 8 + sum(len(t) for t in trajs) * 3 * 8
    roofline = {
        'bound': 'mfma' if is_mfma else 'valu',
        'pipe': ('fp64 matrix pipe (v_mfma_f64_4x4x4_4b)' if is_mfma else
                 'fp64 vector FMA issue (v_fma_f64 / v_fmac_f64_dpp); no MFMA in this kernel'),
        'kernel': kname,
        'achieved': exe / ksec / 1e12, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': exe / ksec / 1e12 / FP64_PEAK_TFLOPS,
        'flops_basis': 'operations the kernel executes: modal recursion on the reduced chain, frames actually run',
        'flop_per_eval_executed': exe / n,
        'frames_executed_fraction': frames_frac,
        'frames_note': 'share of the (candidate, frame) pairs the launch ran itself, counted on the device; the rest comes out '
                       'of the prefix table: a candidate starts at its first switch and, once its filter state agrees with '
                       'the switch-free one behind a switch, takes the table\'s sums up to its next switch',
        'prefix_table': {'bytes': prefix_bytes, 'build_ms_once_per_trajectory_set': prefix_ms,
                         'note': 'switch-free filter states per (trajectory, state, frame): built once per trajectory set by '
                                 'the likelihood kernel itself, outside the timed region like the upload of the trajectory'},
        'kernel_ms': kernel_ms,
        'traffic': traffic,
        'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC passes, gfx950 correction applied; profiles/r02_hbm_traffic.json)',
        'algorithmic_bytes_per_launch': alg_bytes,
        'hbm_algorithmic_GBps': alg_bytes / ksec / 1e9,
        'hbm_frac': alg_bytes / ksec / 1e9 / HBM_PEAK_GBS,
        'attainable_fma_peak': 51.5,
        'attainable_note': 'register-resident v_fma_f64 loop measured on MI355X: 51.5 TFLOP/s at >= 2 waves/SIMD (profiles/r01_f64_rates.txt)',
        'flop_per_eval_canonical': can / n,
        'canonical_equiv_frac': can / ksec / 1e12 / FP64_PEAK_TFLOPS,
        'canonical_note': ('SURVEY 8a canonical F (dense recursion on all N monomers) / kernel time / peak: a SPEED-UP '
                           'equivalent (the kernel runs %d of %d modes in the eigenbasis of B), not a utilisation'
                           % (h.query(_lib.Q_NEFF), h.query(_lib.Q_N))),
    }

    result = {
        'metric': 'logL evaluations/sec (T=1000, 2-state) per AMIS batch',
        'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': workload, 'samples_this_rank': n, 'samples_global': n_global, 'T': T, 'k': k,
                   'states': args.states, 'path': args.path,
                   'collective': ('all_gather(float64[%d]) per step, %s' % (pad, args.backend)) if world > 1 else 'none (1 GPU)'},
        'roofline': roofline,
    }

    # ---- the seam: FixedkSampler.logL(ss, thetas), host arrays in, host array out ----------------------------
    if traj_id is None and not args.no_seam:
        model_for_sampler = bdist.ShardedModel(model) if (world > 1 and args.scaling == 'strong') else model
        sampler = bild_amd.FixedkSampler(trajs[0], model_for_sampler, k=k, N=len(ss), max_fcomplete=0)
        if world > 1 and args.scaling == 'strong':
            # replicated AMIS loop: every rank passes the FULL batch, evaluates its shard, one all-gather
            _, _, ss_full, thetas_full = build_workload(0, n_global, T, k, S=args.states)
            seam_args = (ss_full, thetas_full)
        else:
            seam_args = (ss, thetas)
        got = sampler.logL(*seam_args)
        sdt, skms, _ = timed(lambda: sampler.logL(*seam_args), args.steps, min(args.warmup, 3))
        result['api_seam'] = {
            'what': 'FixedkSampler.logL(ss, thetas): host (N,k+1) float64 + int64 in, host (N,) float64 out, per step '
                    '(bild/amis.py:717-739)',
            'value': (n_global if args.scaling == 'strong' else n * world) * args.steps / sdt, 'unit': 'evals/s',
            'ms_per_call': sdt / args.steps * 1e3, 'kernel_ms': skms,
        }
        step(args.path)
        torch.cuda.synchronize()
        if not (world > 1 and args.scaling == 'strong'):
            result['api_seam']['max_abs_diff_vs_device_entry'] = float(np.max(np.abs(got - d_out[:n].cpu().numpy())))

    if rank == 0 and world == 1 and args.scaling == 'weak' and not args.no_secondary and n == 10000:
        # the throughput regime of the same kernel: twenty times the batch on the same trajectory (the 10k headline is bound
        # by the latency of its longest chain of close switches)
        import helpers as H
        n_big = 200000
        rng_b = np.random.default_rng(4242)
        ss_b, thetas_b = H.candidate_profiles(rng_b, n_big, k, args.states)
        a_b, b_b = segments_from_st(ss_b, thetas_b, T)
        da_b, db_b = torch.from_numpy(a_b).to(dev), torch.from_numpy(b_b).to(dev)
        out_b = torch.empty(n_big, dtype=torch.float64, device=dev)
        order_b = torch.from_numpy(_lib.schedule_segments(h, ts, a_b, b_b, None, path=args.path)).to(dev)

        def big_step():
            _lib.logl_segments_device(h, ts, n_big, k + 1, da_b.data_ptr(), db_b.data_ptr(), 0, out_b.data_ptr(),
                                      stream=torch.cuda.current_stream().cuda_stream, path=args.path, d_order=order_b.data_ptr())
        bdt, bkms, _ = timed(big_step, 10, 2)
        result['large_batch'] = {
            'what': f'{n_big} candidates on the same trajectory, resident in HBM, launch order of bild_schedule_segments',
            'value': n_big * 10 / bdt, 'unit': 'evals/s', 'kernel_ms': bkms, 'frames_executed_fraction': timed.frames_per_launch / (n_big * T),
        }
        del da_b, db_b, out_b, order_b

    if rank == 0 and world == 1 and args.scaling == 'weak' and not args.no_secondary and not args.no_seam:
        # a whole AMIS iteration around the seam (bild/amis.py:805-906): draw N samples from the current
        # proposal (NumPy, the reference's random stream), evaluate them, refit the proposal over ALL samples drawn so far
        np.random.seed(7)
        amis_sampler = bild_amd.FixedkSampler(trajs[0], model, k=k, N=n, max_fev=10 ** 9, max_fcomplete=0)
        t_like = [0.0]
        inner = amis_sampler.logL

        def timed_logl(ss_, thetas_):
            t0_ = time.perf_counter()
            out_ = inner(ss_, thetas_)
            t_like[0] += time.perf_counter() - t0_
            return out_
        amis_sampler.logL = timed_logl
        for _ in range(3):
            amis_sampler.step()
        t_like[0] = 0.0
        amis_steps = 8
        t0 = time.perf_counter()
        for _ in range(amis_steps):
            amis_sampler.step()
        adt = time.perf_counter() - t0
        result['amis_step'] = {
            'what': f'FixedkSampler.step() at N = {n}: draws + likelihood + weights / refit / evidence over the pool '
                    f'({(3 + amis_steps) * n} samples at the end)',
            'ms_per_step': adt / amis_steps * 1e3, 'likelihood_ms_per_step': t_like[0] / amis_steps * 1e3,
            'bookkeeping': 'device (csrc/amis_device.hip)' if getattr(amis_sampler._core, 'on_device', False) else 'host (csrc/amis_host.cpp)',
            'value': n * amis_steps / adt, 'unit': 'samples/s through whole AMIS iterations',
        }

    if rank == 0 and world == 1 and args.scaling == 'weak':
        if not args.no_secondary and prefix_bytes:
            reps = max(5, args.steps // 3)
            ndt, nkms, nname = timed(lambda: step(args.path, prefix=False), reps, 1)
            _, nexe = _lib.flop_count(h, ts, n, traj_id=traj_id, path=args.path)
            result['without_prefix_table'] = {
                'what': 'the same batch with every candidate run frame by frame from frame 0 (BILD_NO_PREFIX, array order)',
                'value': n * reps / ndt, 'unit': 'evals/s', 'kernel': nname, 'kernel_ms': nkms,
                'achieved': nexe / (nkms * 1e-3) / 1e12, 'frac': nexe / (nkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            }
            # the same kernel code in its issue-bound regime, beside the default launch's figure
            roofline['same_kernel_frame_by_frame'] = {'kernel_ms': nkms, 'achieved': nexe / (nkms * 1e-3) / 1e12,
                                                      'frac': nexe / (nkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                                      'note': 'every frame of every candidate run (no prefix table): what the frame '
                                                              'loop itself reaches of the fp64 peak; the default launch runs the '
                                                              'share of the frames given in frames_executed_fraction (the rest comes '
                                                              'out of tables) and lasts as long as its longest chain of close '
                                                              'switches, run by one wavefront (DESIGN.md section 4)'}
            a_out = d_out[:n].cpu().numpy().copy()
            step(args.path)
            torch.cuda.synchronize()
            result['without_prefix_table']['max_abs_diff_vs_default'] = float(np.max(np.abs(a_out - d_out[:n].cpu().numpy())))
        if not args.no_secondary and args.path != 'dense':
            reps = max(3, args.steps // 10)
            ddt, dkms, dname = timed(lambda: step('dense'), reps, 1)
            dcan, dexe = _lib.flop_count(h, ts, n, path='dense')
            result['dense_path'] = {
                'what': 'BILD_PATH_DENSE on the reduced chain (C <- B C B + Sig every frame)',
                'value': n * reps / ddt, 'unit': 'evals/s', 'kernel': dname, 'kernel_ms': dkms,
                'bound': 'mfma' if 'mfma' in dname else 'valu',
                'achieved': dexe / (dkms * 1e-3) / 1e12, 'frac': dexe / (dkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            }
        if not args.no_secondary and not args.no_reduce:
            # the reference's literal algorithm: all N monomers, C <- B C B + Sig every frame
            # (canonical flop count == executed flop count)
            a_ = model.arrays()
            h_full = _lib.ModelHandle(a_['B'], a_['G'], a_['Sig'], a_['M0'], a_['C0'], model.measurement, reduce=False)
            ts_full = _lib.TrajSetHandle(h_full, [np.asarray(trajs[0][:])], np.asarray(model.localization_error)[None, :])

            def canon():
                _lib.logl_segments_device(h_full, ts_full, n, k + 1, d_start.data_ptr(), d_state.data_ptr(), 0,
                                          d_out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, path='dense')
            cdt, ckms, cname = timed(canon, 3, 1)
            ccan, cexe = _lib.flop_count(h_full, ts_full, n, path='dense')
            full_out = d_out[:n].cpu().numpy().copy()
            result['canonical_path'] = {
                'what': 'dense path without reduction: the reference recursion itself on all %d monomers' % h_full.query(_lib.Q_N),
                'value': n * 3 / cdt, 'unit': 'evals/s', 'kernel': cname, 'kernel_ms': ckms, 'bound': 'mfma' if 'mfma' in cname else 'valu',
                'achieved': cexe / (ckms * 1e-3) / 1e12, 'unit_achieved': 'TFLOP/s fp64',
                'frac': cexe / (ckms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            }
            step(args.path)
            torch.cuda.synchronize()
            result['canonical_path']['max_abs_diff_vs_default_path'] = float(np.max(np.abs(full_out - d_out[:n].cpu().numpy())))
        if not args.no_cpu_baseline:
            base, ref_out = cpu_baseline(rank, n, T, k, args.states)
            result['cpu_baseline'] = base
            step(args.path)
            torch.cuda.synchronize()
            got = d_out[:len(ref_out)].cpu().numpy()
            result['parity_max_abs_diff_vs_cpu_baseline'] = float(np.max(np.abs(got - ref_out)))
            result['speedup_vs_cpu_baseline'] = value / base['value']
            if args.cpu_allcores:
                cores = _cpu_share()
                allc, _ = cpu_baseline(rank, n, T, k, args.states, budget_s=10.0, procs=cores)
                result['cpu_baseline_allcores'] = allc
                result['speedup_vs_cpu_allcores'] = value / allc['value']

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-baseline-child':
        _cpu_baseline_child(sys.argv[2:])
    else:
        main()
