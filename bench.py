#!/usr/bin/env python3
"""
Headline benchmark: logL evaluations / second of one AMIS batch on MI355X.

Workload (BASELINE.json configs[1]): a batch of 10 000 candidate looping profiles x 1 trajectory, T = 1000 frames,
2-state Rouse model, N = 20 monomers, d = 3, one localization error (d* = 1), k = 4 switches per candidate, fp64.
One "step" = one pass of the hot path over one such batch: everything `FixedkSampler.logL(ss, thetas)` does for one
AMIS iteration (reference bild/amis.py:717-739).

What the JSON line reports
  value / ms_per_step   the batch as the sampler produced it -- (s, theta) rows, float64 + one byte per state -- resident
                        in HBM, a DIFFERENT batch every step (eight of them rotate), through `bild_logl_st_device`: switch
                        frames (st2profile), cleaning, table walk, frame loop, all inside the timed region; nothing is
                        converted, ordered or scheduled beforehand and nothing crosses PCIe (the bench contract's definition);
  api_seam              the same batches through the seam the path is defined by: `FixedkSampler.logL(ss, thetas)` with
                        host arrays in and a host array out per step (one staged H2D copy, the same kernels, one D2H copy);
  roofline              the dominant kernel (the frame loop over the work lists) against the fp64 vector peak, from its
                        average launch duration measured with HIP events on the launch stream; the table-walk kernel beside it;
  k_sweep, large_batch, config2_one_gpu, config3, config4, single_eval, concentrated_proposal, first_call
                        the other BASELINE configurations and the regimes a caller meets, each with its own timing;
  amis_step             whole AMIS iterations around the seam: NumPy random stream (the reference's), and draws on the device;
  cpu_baseline          the oracle's C port of the reference kernel on one host core (+ reference-equivalent figure).

Multi-GPU: one process per GPU.  `--scaling weak` (default, what the driver runs): every rank evaluates its own 10k
batch and joins ONE all-gather of the log-likelihoods per step (RCCL), as an AMIS step needs them to form the
importance weights.  `--scaling strong`: a FIXED workload is partitioned over the ranks -- `--strong-config 1`:
configs[1] (10k x 1 trajectory, contiguous sample shards), `--strong-config 2`: configs[2] (256 trajectories x 1000
samples, whole trajectories per rank, `dist.shard_by_trajectory`) -- again one all-gather per step.  Every multi-GPU line
carries `multi_gpu`: per-rank kernel and collective times, the rank count RCCL saw, and the builder's projection.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
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
ROTATE = 8                # distinct batches resident in HBM, one per step in turn


def kernel_source_hash():
    """ what the measured HBM traffic (profiles/r04_hbm_traffic.json) belongs to: the kernel sources it was measured on """
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'bild_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.h', '.cpp')) and name != 'asan_stubs.cpp':     # (what goes into libbild_amd.so)
            with open(os.path.join(csrc, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def build_workload(seed, n_samples, T, k, S=2, N=20, d=3, err=0.1, n_traj=1, batches=1):
    """ model, n_traj trajectories, and `batches` independent batches of n_samples candidates per trajectory """
    import helpers as H
    import bild_amd
    rng_t = np.random.default_rng(1000 + seed)
    model = bild_amd.MultiStateRouse(N, 1., 5., d=d, looppositions=H.LOOPS[S], localization_error=err)
    trajs = []
    for _ in range(n_traj):
        truth = H.random_profile(rng_t, T, S, T // 5)
        trajs.append(model.trajectory_from_loopingprofile(truth, rng=rng_t))
    out = []
    for b in range(batches):
        rng_c = np.random.default_rng(2000 + seed + 7919 * b)
        out.append(H.candidate_profiles(rng_c, n_samples * n_traj, k, S))
    if batches == 1:
        return model, trajs, out[0][0], out[0][1]
    return model, trajs, out


def _cpu_baseline_loop(model, traj, ss, thetas, T, budget_s, first):
    """ the timed loop itself; runs in a child process (see cpu_baseline) """
    import helpers as H
    from oracle import oracle
    # The reference's Cython kernel does not travel to the GPU box (oracle/_ref/ is in .gpurunignore): what is
    # timed here, on every box, is this repository's C restatement of it (oracle/msrouse_logl.c, bit-pinned to the reference
    # goldens), in the Python loop FixedkSampler.logL runs (amis.py:735-739).
    states = H.expand(ss[first:first + 8192], thetas[first:first + 8192], T)
    arrays, w, err, x = model.arrays(), model.measurement, model.localization_error, traj[:]

    def ref(prof):
        return oracle.logl(arrays, w, err, x, prof)
    ref(states[0])  # warm
    out = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and len(out) < len(states):
        out.append(ref(states[len(out)]))
    return 'port', time.perf_counter() - t0, np.array(out)


def _cpu_baseline_child(argv):
    """ `bench.py --cpu-baseline-child seed n T k S budget out.npz first`: CPU only, never touches the GPU """
    seed, n, T, k, S = (int(v) for v in argv[:5])
    model, trajs, ss, thetas = build_workload(seed, n, T, k, S=S)
    kind, dt, out = _cpu_baseline_loop(model, trajs[0], ss, thetas, T, float(argv[5]), int(argv[7]))
    np.savez(argv[6], kind=kind, dt=dt, out=out)


def cpu_baseline(seed, n, T, k, S, budget_s=15.0, procs=1):
    """
    The oracle's C restatement of the reference kernel (kind "port") on host cores, driven exactly like FixedkSampler.logL
    drives the reference's (a Python loop, amis.py:735-739), on a bounded sample of the same batch; `reference_equiv`
    converts it with the factor oracle/conversion_factor.py measured in the build container.  Every
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
    base = dict(value=total / max(dts), unit='evals/s', cores=procs, kind=kind,
                sample=f"{total} profiles of the rank-0 batch, T={T}, {max(dts):.1f} s of the oracle's C restatement of the "
                       f"reference kernel (oracle/msrouse_logl.c) in a Python loop (amis.py:735-739), {procs} process(es)")
    try:
        # reference Cython : port, measured in the BUILD CONTAINER on one core by oracle/conversion_factor.py (the reference
        # never leaves that container: SURVEY 8d)
        with open(os.path.join(ROOT, 'oracle', 'conversion_factor.json')) as f:
            cf = json.load(f)
        base['reference_equiv'] = {
            'value': base['value'] * cf['reference_over_port'], 'unit': 'evals/s', 'factor': cf['reference_over_port'],
            'provenance': f"{cf['script']} on {cf['measured']}: reference Cython {cf['reference_evals_per_s']:.1f} evals/s vs port "
                          f"{cf['port_evals_per_s']:.1f} evals/s on one core of the build container ({cf['host']['cpu']}), same inputs; "
                          "on the GPU box's EPYC host the two were 288.5 vs 243.8 evals/s = 1.18 in round 1, when the "
                          "binary still travelled (profiles/r01_cpu_allcores.txt)"}
    except Exception:
        base['reference_equiv'] = None
    return base, outs[0]


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
    ap.add_argument('--repeats', type=int, default=60, help='timed regions of exactly --steps steps; the median one is reported')
    ap.add_argument('--sustained', type=float, default=4.0, help='seconds of back-to-back steps for the `sustained` figure (0: skip)')
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
    ap.add_argument('--no-secondary', action='store_true', help='skip every side measurement (profiling runs that want the headline launches alone)')
    ap.add_argument('--no-seam', action='store_true', help='skip the api_seam measurement')
    ap.add_argument('--sweep', action='store_true', help='also the N x d* sweep of SURVEY 8(d) (chain lengths 4 ... 32, one and two localization errors)')
    ap.add_argument('--exchange', default='auto', choices=['auto', 'rccl', 'direct'],
                    help="the step's collective: 'rccl' = all_gather_into_tensor (ring), 'direct' = the library's one-shot peer "
                         "write (bild_exchange_*: every rank stores its shard into every peer's receive block, one kernel), "
                         "'auto' = direct where a self-test of it passes on every rank (it has never run between two GPUs in "
                         "the builder's pool), else rccl")
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
    import helpers as H
    from bild_amd import _lib
    from bild_amd import dist as bdist

    T, k = args.T, args.k
    dev = torch.device('cuda', dev_index)
    stream_of = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

    # ---- the workload of this rank ------------------------------------------------------------------------
    if args.scaling == 'weak':
        n = args.samples
        model, trajs, batches = build_workload(rank, n, T, k, S=args.states, batches=ROTATE)
        traj_id = None
        n_global = n * world
        sizes = [n] * world
        workload = (f'configs[1]: {n} profile samples x 1 trajectory per GPU, T={T}, {args.states}-state Rouse N=20 d=3 '
                    f'd*=1, k={k} switches, fp64; {ROTATE} distinct batches resident in HBM as (s, theta) rows, one per step in turn')
    elif args.strong_config == 1:
        n_global = args.samples
        model, trajs, full = build_workload(0, n_global, T, k, S=args.states, batches=ROTATE)
        lo, hi = bdist.shard_bounds(n_global, world, rank)
        batches = [(ss_[lo:hi], th_[lo:hi]) for ss_, th_ in full]
        traj_id = None
        n = hi - lo
        sizes = [b - a for a, b in (bdist.shard_bounds(n_global, world, r) for r in range(world))]
        workload = (f'configs[1] strong: {n_global} profile samples x 1 trajectory split over {world} GPU(s), T={T}, '
                    f'{args.states}-state, k={k}, fp64')
    else:
        n_traj_total, per = 256, 1000
        model, trajs_all, full = build_workload(0, per, T, k, S=args.states, n_traj=n_traj_total, batches=2)
        owners = bdist.shard_by_trajectory([T] * n_traj_total, [per] * n_traj_total, world)
        mine = owners[rank]
        trajs = [trajs_all[j] for j in mine]
        rows = np.concatenate([np.arange(j * per, (j + 1) * per) for j in mine])
        batches = [(ss_[rows], th_[rows]) for ss_, th_ in full]
        traj_id = np.repeat(np.arange(len(mine)), per).astype(np.int32)
        n = len(rows)
        n_global = n_traj_total * per
        sizes = [len(o) * per for o in owners]
        workload = (f'configs[2] strong: {n_traj_total} trajectories x {per} samples sharded by trajectory over {world} '
                    f'GPU(s) ({len(mine)} trajectories on this rank), T={T}, {args.states}-state, k={k}, fp64')
    ss, thetas = batches[0]

    model.path = args.path
    if args.no_reduce:
        a_ = model.arrays()
        model._handle = _lib.ModelHandle(a_['B'], a_['G'], a_['Sig'], a_['M0'], a_['C0'], model.measurement, reduce=False)
    h = model.handle()
    ts = model.trajset(trajs if traj_id is not None else trajs[0])      # trajectories resident in HBM
    d_batches = [(torch.from_numpy(np.ascontiguousarray(ss_)).to(dev), torch.from_numpy(th_.astype(np.uint8)).to(dev))
                 for ss_, th_ in batches]                                 # candidates resident in HBM, as the sampler produced them
    d_tid = torch.from_numpy(traj_id).to(dev) if traj_id is not None else None
    _lib.logl_st(h, ts, ss, thetas, traj_id, path=args.path)             # one evaluation: the set's tables exist now
    pad = max(sizes)
    d_out = torch.zeros(pad, dtype=torch.float64, device=dev)
    d_all = torch.empty(pad * world, dtype=torch.float64, device=dev)
    turn = [0]
    coll_events = []
    direct, exchange_note = None, None
    if world > 1 and (args.exchange == 'direct' or (args.exchange == 'auto' and args.backend == 'nccl')):
        # (with --backend gloo the ranks may share a GPU: a rehearsal of this very path on a box with fewer GPUs than ranks)
        # self-test: three exchanges of recognisable shards with a short timeout, verdict agreed between the ranks
        ok = 1
        try:
            direct = bdist.DirectExchange.from_torch(pad)
            direct._x.set_timeout(2.0)
            probe = torch.empty(pad, dtype=torch.float64, device=dev)
            for trial in range(3):
                probe.fill_(float(1000 * trial + rank))
                direct.allgather(probe.data_ptr(), d_all.data_ptr(), pad, stream_of())
                torch.cuda.synchronize()
                direct.status()
                got = d_all.view(world, pad)[:, ::max(pad // 7, 1)].cpu().numpy()
                ok &= int(all(np.all(got[r] == 1000 * trial + r) for r in range(world)))
            direct._x.set_timeout(5.0)
        except Exception as exc:   # (IPC mapping refused, a peer that never delivered ...)
            ok, exchange_note = 0, repr(exc)
        verdict = torch.tensor([ok], dtype=torch.int32, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        if int(verdict.item()) != 1:
            if args.exchange == 'direct':
                raise SystemExit(f"--exchange direct: the self-test failed on some rank ({exchange_note})")
            direct, exchange_note = None, f"direct exchange self-test failed on some rank ({exchange_note}): RCCL all-gather used"

    def step(path=None, **kw):
        d_ss, d_th = d_batches[turn[0] % len(d_batches)]
        turn[0] += 1
        _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), d_out.data_ptr(),
                            d_traj_id=d_tid.data_ptr() if d_tid is not None else 0, stream=stream_of(), path=path or args.path, **kw)
        if world > 1:                                 # the one collective of an AMIS step
            if args.backend == 'nccl' or direct is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if direct is not None:
                    direct.allgather(d_out.data_ptr(), d_all.data_ptr(), pad, stream_of())
                else:
                    bdist.all_gather_logl(d_out, d_all)
                e1.record()
                coll_events.append((e0, e1))
            else:
                d_all.copy_(bdist.all_gather_logl(d_out.cpu()))

    def timed(fn, steps, warmup, sample=4, handle=None, repeats=1):
        # The timed region -- EXACTLY `steps` steps between barrier + synchronize on both sides, maximum over ranks -- is run
        # `repeats` times and the MEDIAN region is what is reported: one region of a latency-bound launch is a millisecond
        # long and scatters by +-5 % from run to run (round 3: 168.6 M in the driver's run, 172-178 M in the builder's).
        # (kernel durations from HIP events around every `sample`-th launch of the timed regions: the events cost a few
        # microseconds per launch, which every step would otherwise pay)
        for _ in range(warmup):
            fn()
        coll_events.clear()
        _lib.kernel_timing(sample if steps >= 2 * sample else 1)
        dts = []
        for _ in range(repeats):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dts.append(time.perf_counter() - t0)
        _lib.kernel_timing(False)
        kms, launches, kname = _lib.kernel_timing_read()
        wms, wl = _lib.kernel_timing_read_walk()
        timed.frames_per_launch = _lib.frames_run_read(handle or h) / max(launches, 1)   # counted on the device by the tasks themselves
        timed.walk_ms = wms / max(wl, 1)
        dts = np.array(dts)
        timed.local_dt = float(np.median(dts))
        if world > 1:
            t = torch.from_numpy(dts).to(dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = t.cpu().numpy()
        q25, med, q75 = (float(v) for v in np.percentile(dts, [25, 50, 75]))
        per = 1e3 / steps
        timed.stats = {'repeats': int(repeats), 'median': med * per, 'q25': q25 * per, 'q75': q75 * per, 'iqr': (q75 - q25) * per,
                       'min': float(dts.min()) * per, 'max': float(dts.max()) * per, 'timed_region_s': float(dts.sum()),
                       'what': f'ms per step over {repeats} timed regions of exactly {steps} steps each (barrier + synchronize on both '
                               'sides, maximum over ranks per region); `value` and `ms_per_step` are the median region'}
        return med, kms / max(launches, 1), kname

    def sustained(fn, seconds, per_step_s, chunk=512):
        """
        back-to-back steps for about `seconds` (one synchronisation per `chunk` steps): -> (steps per second, wall).  The number
        of steps is fixed beforehand from the measured step time (the same on every rank: the steps hold a collective)
        """
        chunks = max(1, int(np.ceil(seconds / max(per_step_s, 1e-6) / chunk)))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(chunks):
            for _ in range(chunk):
                fn()
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        return chunks * chunk / wall, wall

    dt, kernel_ms, kname = timed(step, args.steps, args.warmup, repeats=args.repeats)
    walk_ms = timed.walk_ms
    headline_stats = timed.stats
    value = n_global * args.steps / dt
    frames_total = float(sum(len(trajs[j]) for j in (traj_id if traj_id is not None else np.zeros(n, dtype=int))))
    frames_frac = timed.frames_per_launch / frames_total

    # ---- roofline of the dominant kernel: executed operations against the fp64 vector peak --------------------
    traffic, traffic_note = None, 'no measurement for this workload / these kernel sources'
    try:
        with open(os.path.join(ROOT, 'profiles', 'r04_hbm_traffic.json')) as f:
            tj = json.load(f)
        if tj['workload'] == {'samples': n, 'T': T, 'k': k, 'path': args.path} and args.states == 2:
            if tj.get('kernel_source_hash') == kernel_source_hash():
                traffic = tj['traffic_bytes_corrected']
                traffic_note = 'HBM bytes per step (both kernels), rocprofv3 PMC passes, gfx950 correction applied (profiles/r04_hbm_traffic.json)'
            else:
                traffic_note = ('profiles/r04_hbm_traffic.json was measured on other kernel sources (hash %s, now %s): not quoted'
                                % (tj.get('kernel_source_hash'), kernel_source_hash()))
    except Exception:
        pass
    can, exe = _lib.flop_count(h, ts, n, traj_id=traj_id, path=args.path)
    exe *= frames_frac
    prefix_bytes, prefix_ms = _lib.prefix_info(ts)
    ksec = kernel_ms * 1e-3
    is_mfma = 'mfma' in kname
    alg_bytes = n * (k + 1) * 9 + n * 8 + sum(len(t) for t in trajs) * 3 * 8
    table_bytes = n * k * 64          # per switch: a 16-byte entry, two running sums, at most one pair entry with two more
    roofline = {
        'bound': 'mfma' if is_mfma else 'valu',
        'pipe': ('fp64 matrix pipe (v_mfma_f64_4x4x4_4b)' if is_mfma else
                 'fp64 vector FMA issue (v_fma_f64 / v_fmac_f64_dpp); no MFMA in this kernel'),
        'kernel': kname + ' over the work lists (the frame loop: chains of three and more close switches)',
        'achieved': exe / ksec / 1e12, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': exe / ksec / 1e12 / FP64_PEAK_TFLOPS,
        'flops_basis': 'operations the kernel executes: modal recursion on the reduced chain, frames actually run',
        'flop_per_eval_executed': exe / n,
        'frames_executed_fraction': frames_frac,
        'frames_note': 'share of the (candidate, frame) pairs the launch ran itself, counted on the device; the rest comes out '
                       'of tables: the table-walk kernel (one lane per candidate) finishes every candidate whose switches are '
                       'covered by the transient / pair tables and hands the others to the frame loop, where a chain of close '
                       'switches starts at its second switch from the transient state table and ends at the first frame at which '
                       'its filter state agrees with the switch-free one',
        'kernel_ms': kernel_ms,
        'walk_kernel': {'kernel': 'walk_kernel (csrc/walk.hip): (s, theta) -> switch frames, cleaning, table walk; one lane per candidate',
                        'kernel_ms': walk_ms,
                        'hbm_algorithmic_GBps': (alg_bytes + table_bytes) / max(walk_ms * 1e-3, 1e-12) / 1e9,
                        'hbm_frac': (alg_bytes + table_bytes) / max(walk_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS},
        'latency_note': 'neither roof binds: the step is the walk kernel (one memory round trip per group of switches) plus the '
                        'longest chain of close switches, run frame by frame by ONE wavefront (DESIGN.md section 4)',
        'tables': {'bytes': prefix_bytes, 'build_ms_device': prefix_ms,
                   'note': 'prefix / transient / transient-state / pair tables of the trajectory set, built once by the likelihood '
                           'kernel itself at the first evaluation.  build_ms_device is the FIRST set of the process: its builder '
                           'launches include loading their code objects (once per process); `first_call` below times a second '
                           'set, warm'},
        'traffic': traffic,
        'traffic_note': traffic_note,
        'algorithmic_bytes_per_launch': alg_bytes + table_bytes,
        'algorithmic_bytes_note': '(s, theta) rows 9 B per segment, results 8 B, trajectory 24 KB, + 64 B of table entries per switch',
        'hbm_algorithmic_GBps': (alg_bytes + table_bytes) / ((kernel_ms + walk_ms) * 1e-3) / 1e9,
        'hbm_frac': (alg_bytes + table_bytes) / ((kernel_ms + walk_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        'attainable_fma_peak': 51.5,
        'attainable_note': 'register-resident v_fma_f64 loop measured on MI355X: 51.5 TFLOP/s at >= 2 waves/SIMD (profiles/r01_f64_rates.txt)',
        'flop_per_eval_canonical': can / n,
        'canonical_equiv_frac': can / ((kernel_ms + walk_ms) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
        'canonical_note': ('SURVEY 8a canonical F (dense recursion on all N monomers) / device time of a step / peak: a SPEED-UP '
                           'equivalent (the kernels run %d of %d modes in the eigenbasis of B and 0.3 %% of the frames), not a utilisation'
                           % (h.query(_lib.Q_NEFF), h.query(_lib.Q_N))),
    }

    result = {
        'metric': 'logL evaluations/sec (T=1000, 2-state) per AMIS batch',
        'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
        'ms_per_step_stats': headline_stats, 'timed_region_s': headline_stats['timed_region_s'],
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': workload, 'samples_this_rank': n, 'samples_global': n_global, 'T': T, 'k': k,
                   'states': args.states, 'path': args.path,
                   'entry': 'bild_logl_st_device: (s, theta) rows resident in HBM -> log-likelihoods in HBM',
                   'collective': ('all_gather(float64[%d]) per step, %s' % (pad, 'direct exchange (bild_exchange_allgather)' if direct is not None else args.backend)) if world > 1 else 'none (1 GPU)'},
        'roofline': roofline,
    }

    if args.sustained > 0:
        rate, wall_s = sustained(step, args.sustained, dt / args.steps)
        result['sustained'] = {'what': 'back-to-back steps for at least %.0f s, one synchronisation per 512 steps (no barrier inside: every '
                                       'rank runs its own loop)' % args.sustained,
                               'value': n_global * rate, 'unit': 'evals/s', 'ms_per_step': 1e3 / rate, 'wall_s': wall_s}
    if world > 1:
        # ---- what a scaling run needs to explain itself -----------------------------------------------------------------
        coll_ms = float(np.mean([a.elapsed_time(b) for a, b in coll_events])) if coll_events else None
        info = torch.tensor([kernel_ms + walk_ms, coll_ms if coll_ms is not None else -1.0, timed.local_dt / args.steps * 1e3],
                            dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        gathered = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(gathered, info)
        per_rank = [[float(v) for v in g.tolist()] for g in gathered]
        result['multi_gpu'] = {
            'ranks_seen': dist.get_world_size(), 'backend': args.backend,
            'exchange': 'direct (bild_exchange_allgather: one-shot peer writes over IPC-mapped receive blocks)' if direct is not None else 'rccl all_gather_into_tensor',
            'exchange_note': exchange_note,
            'per_rank_kernel_ms': [p[0] for p in per_rank],
            'per_rank_collective_ms': [p[1] if p[1] >= 0 else None for p in per_rank],
            'per_rank_ms_per_step': [p[2] for p in per_rank],
            'collective': 'float64[%d] per rank gathered on the kernels\' stream, events around the call' % pad,
            'projection': 'profiles/r04_scaling_projection.txt: weak scaling = the one-GPU step (57 us) + the collective: direct '
                          'exchange 8.6 us behind the kernels (measured on one GPU) + the xGMI hop and arrival skew (4-7 us, not '
                          'measured) -> ~71 us, 6.4-6.6 x at N = 8; RCCL ring all-gather 30-50 us -> 4.2-5.2 x',
        }

    if rank == 0 and world == 1:
        result['parity_note'] = 'the timed batch is checked below against the oracle (C restatement of the reference Cython kernel, pinned to the reference goldens): parity_max_abs_diff_vs_cpu_baseline'

    # ---- the seam: FixedkSampler.logL(ss, thetas), host arrays in, host array out ----------------------------
    if traj_id is None and not args.no_seam:
        model_for_sampler = bdist.ShardedModel(model) if (world > 1 and args.scaling == 'strong') else model
        sampler = bild_amd.FixedkSampler(trajs[0], model_for_sampler, k=k, N=len(ss), max_fcomplete=0)
        if world > 1 and args.scaling == 'strong':
            # replicated AMIS loop: every rank passes the FULL batch, evaluates its shard, one all-gather
            seam_batches = [build_workload(0, n_global, T, k, S=args.states)[2:4]]
        else:
            seam_batches = batches
        sturn = [0]

        def seam_step():
            ss_, th_ = seam_batches[sturn[0] % len(seam_batches)]
            sturn[0] += 1
            return sampler.logL(ss_, th_)
        got = sampler.logL(*seam_batches[0])
        sdt, skms, _ = timed(seam_step, args.steps, min(args.warmup, 3), repeats=args.repeats)
        result['api_seam'] = {
            'what': 'FixedkSampler.logL(ss, thetas): host (N,k+1) float64 + int64 in, host (N,) float64 out, per step '
                    '(bild/amis.py:717-739); the rows go up as they are (float64 + one byte per state) and are converted on the device',
            'value': (n_global if args.scaling == 'strong' else n * world) * args.steps / sdt, 'unit': 'evals/s',
            'ms_per_call': sdt / args.steps * 1e3, 'kernel_ms': skms, 'walk_kernel_ms': timed.walk_ms,
            'ms_per_call_stats': timed.stats,
        }
        if args.sustained > 0 and world == 1:
            rate, wall_s = sustained(seam_step, args.sustained / 2, sdt / args.steps)
            result['api_seam']['sustained'] = {'value': n * rate, 'unit': 'evals/s', 'ms_per_call': 1e3 / rate, 'wall_s': wall_s}
        if not (world > 1 and args.scaling == 'strong'):
            turn[0] = 0
            step()
            torch.cuda.synchronize()
            result['api_seam']['max_abs_diff_vs_device_entry'] = float(np.max(np.abs(got - d_out[:n].cpu().numpy())))

    secondary = rank == 0 and world == 1 and args.scaling == 'weak' and not args.no_secondary

    def resident(ss_, th_, tid_=None):
        return (torch.from_numpy(np.ascontiguousarray(ss_)).to(dev), torch.from_numpy(th_.astype(np.uint8)).to(dev),
                torch.from_numpy(tid_).to(dev) if tid_ is not None else None)

    def time_resident(handle, tset, d_ss, d_th, d_tid_, count, k1, reps, warm=2, **kw):
        out_ = torch.empty(count, dtype=torch.float64, device=dev)

        def go():
            _lib.logl_st_device(handle, tset, count, k1, d_ss.data_ptr(), d_th.data_ptr(), out_.data_ptr(),
                                d_traj_id=d_tid_.data_ptr() if d_tid_ is not None else 0, stream=stream_of(), **kw)
        dt_, kms_, _ = timed(go, reps, warm, handle=handle)
        return dt_ / reps, kms_, timed.walk_ms, timed.frames_per_launch, out_

    if secondary:
        # ---- how the step moves with the number of switches per candidate ---------------------------------------------
        sweep = {}
        for kk in (2, 4, 8, 15):
            rng_k = np.random.default_rng(3000 + kk)
            ss_k, th_k = H.candidate_profiles(rng_k, n, kk, args.states)
            d1, d2, _ = resident(ss_k, th_k)
            per, kms_, wms_, fr_, _ = time_resident(h, ts, d1, d2, None, n, kk + 1, 30)
            sweep[f'k={kk}'] = {'evals_per_s': n / per, 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_, 'walk_kernel_ms': wms_,
                                'frames_run_per_candidate': fr_ / n}
        result['k_sweep'] = {'what': f'{n} candidates with k uniformly placed switches each (Dirichlet(1) interval lengths), resident in '
                                     'HBM; k = 4 is the headline', **sweep}

        # ---- the throughput regime of the same kernels: twenty times the batch on the same trajectory ------------------
        n_big = 200000
        rng_b = np.random.default_rng(4242)
        ss_b, th_b = H.candidate_profiles(rng_b, n_big, k, args.states)
        d1, d2, _ = resident(ss_b, th_b)
        per, kms_, wms_, fr_, _ = time_resident(h, ts, d1, d2, None, n_big, k + 1, 10)
        result['large_batch'] = {
            'what': f'{n_big} candidates on the same trajectory, (s, theta) rows resident in HBM',
            'value': n_big / per, 'unit': 'evals/s', 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_, 'walk_kernel_ms': wms_,
            'frames_executed_fraction': fr_ / (n_big * T)}
        del d1, d2

        # ---- late AMIS: candidates drawn from a proposal that has concentrated on the true switches -------------------
        truth = np.asarray(trajs[0].meta['loopingprofile'])
        sw = np.nonzero(np.diff(truth))[0] + 1
        kc = int(min(len(sw), 8))
        if kc >= 1:
            pick = np.sort(np.random.default_rng(5).choice(sw, size=kc, replace=False))
            mean = np.diff(np.concatenate([[0], pick, [T]])) / T
            rng_c = np.random.default_rng(6)
            ss_c = rng_c.dirichlet(mean * 2000.0, size=n)               # switches within a few frames of the true ones
            th_c = np.empty((n, kc + 1), dtype=np.int64)
            th_c[:, 0] = truth[0]
            for i in range(1, kc + 1):
                th_c[:, i] = truth[min(pick[i - 1], T - 1)]
            d1, d2, _ = resident(ss_c, th_c)
            per, kms_, wms_, fr_, _ = time_resident(h, ts, d1, d2, None, n, kc + 1, 30)
            result['concentrated_proposal'] = {
                'what': f'{n} candidates with k = {kc} switches drawn from Dirichlet(2000 x true interval lengths): the late-AMIS regime, '
                        'where the candidates agree on the switches to within a few frames',
                'evals_per_s': n / per, 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_, 'walk_kernel_ms': wms_,
                'frames_run_per_candidate': fr_ / n}

        # ---- what the reference's own plug points get: one evaluation per call -----------------------------------------
        prof = H.random_profile(np.random.default_rng(8), T, args.states, T // 5)
        for _ in range(3):
            model.logL(prof, trajs[0])
        t0 = time.perf_counter()
        for _ in range(200):
            model.logL(prof, trajs[0])
        single = (time.perf_counter() - t0) / 200
        lat = {}
        for cnt in (9, 41):
            profs = np.stack([H.random_profile(np.random.default_rng(100 + i), T, args.states, T // 5) for i in range(cnt)])
            for _ in range(3):
                model.logL_batch(profs, trajs[0])
            t0 = time.perf_counter()
            for _ in range(100):
                model.logL_batch(profs, trajs[0])
            per_call = (time.perf_counter() - t0) / 100
            lat[f'{cnt}_profiles'] = {'ms_per_call': per_call * 1e3, 'evals_per_s': cnt / per_call}
        result['single_eval'] = {
            'what': 'model.logL(profile, traj): ONE expanded profile per call, host to host -- what an unchanged reference gets through '
                    'the symbol swap of bild/cython_imports.py:3-7 (and the per-sample hook amis.py:734-736); 9 / 41 profiles per call: '
                    'the batches of postproc.logLR_boundaries (bild/postproc.py:36-59, 2k+1 profiles)',
            'ms_per_call': single * 1e3, 'evals_per_s': 1.0 / single, **lat}

        # ---- the first evaluation on a fresh trajectory set (all code objects loaded by now): what the tables cost -------
        rng_f = np.random.default_rng(77)
        tr2 = model.trajectory_from_loopingprofile(H.random_profile(rng_f, T, args.states, T // 5), rng=rng_f)
        t0 = time.perf_counter()
        ts2 = model.trajset(tr2)
        t1 = time.perf_counter()
        _lib.logl_st(h, ts2, ss[:100], thetas[:100], path=args.path)
        t2 = time.perf_counter()
        _lib.logl_st(h, ts2, ss[:100], thetas[:100], path=args.path)
        t3 = time.perf_counter()
        b2, ms2 = _lib.prefix_info(ts2)
        result['first_call'] = {
            'what': 'a second trajectory of the same length in the same process: upload, first evaluation of 100 candidates (builds the '
                    'prefix / transient / state / pair tables), second evaluation of the same 100',
            'upload_ms': (t1 - t0) * 1e3, 'first_evaluation_ms': (t2 - t1) * 1e3, 'second_evaluation_ms': (t3 - t2) * 1e3,
            'tables_bytes': b2, 'tables_build_ms_device': ms2}

    if secondary and not args.no_seam:
        # ---- whole AMIS iterations around the seam (bild/amis.py:805-906) ----------------------------
        amis = {}
        for mode, kw in (('numpy_stream', {}), ('device_rng', {'rng': 'device', 'seed': 7})):
            np.random.seed(7)
            smp = bild_amd.FixedkSampler(trajs[0], model, k=k, N=n, max_fev=10 ** 9, max_fcomplete=0, **kw)
            for _ in range(3):
                smp.step()
            amis_steps = 10
            t0 = time.perf_counter()
            for _ in range(amis_steps):
                smp.step()
            adt = time.perf_counter() - t0
            amis[mode] = {'ms_per_step': adt / amis_steps * 1e3, 'samples_per_s': n * amis_steps / adt,
                          'fused': bool(smp._fusable()), 'pool_at_the_end': len(smp._core),
                          'evidence': [float(v) for v in smp.evidences[-1]]}
        result['amis_step'] = {
            'what': f'FixedkSampler.step() at N = {n}: draws + likelihood + weights / refit / evidence over the pool, one native call per '
                    'step (bild_amis_step_fused: the samples go up once, likelihood and the three passes over the pool on one stream). '
                    'numpy_stream: the reference\'s random stream (NumPy draws on the host, 1.0 ms of the step); device_rng: opt-in, '
                    'the draws on the GPU (Philox-4x32-10, not the reference\'s random numbers)',
            **amis}

    if secondary:
        # ---- BASELINE configs[2] on one GPU: 256 trajectories x 1000 candidates ----------------------------------------
        try:
            n_traj2, per2 = 256, 1000
            model2, trajs2, _, _ = build_workload(0, 1, T, k, S=args.states, n_traj=n_traj2)
            rng2 = np.random.default_rng(99)
            ss2, th2 = H.candidate_profiles(rng2, n_traj2 * per2, k, args.states)
            tid2 = np.repeat(np.arange(n_traj2), per2).astype(np.int32)
            t0 = time.perf_counter()
            ts_2 = model2.trajset(trajs2)
            _lib.logl_st(model2.handle(), ts_2, ss2[:1000], th2[:1000], tid2[:1000])
            build_s = time.perf_counter() - t0
            d1, d2, d3 = resident(ss2, th2, tid2)
            per, kms_, wms_, fr_, _ = time_resident(model2.handle(), ts_2, d1, d2, d3, n_traj2 * per2, k + 1, 10)
            b2, ms2 = _lib.prefix_info(ts_2)
            result['config2_one_gpu'] = {
                'what': f'BASELINE configs[2] on ONE GPU: {n_traj2} trajectories x {per2} candidates, T={T}, k={k}, (s, theta) rows and '
                        'traj_id resident in HBM',
                'value': n_traj2 * per2 / per, 'unit': 'evals/s', 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_,
                'walk_kernel_ms': wms_, 'frames_executed_fraction': fr_ / (n_traj2 * per2 * T),
                'tables_bytes': b2, 'tables_build_ms_device': ms2, 'upload_and_first_evaluation_s': build_s,
                'tables_note': 'prefix + transient + pair tables; the transient STATE table (26 GB for this set, 0.8 s to allocate and '
                               'fill) is built for sets of up to 4 GB of it only, unless the caller declares >= 1e8 evaluations '
                               '(bild_trajset_expect) or sets BILD_STATES_MAX_BYTES'}
            # the same set declared for a long life (bild_trajset_expect >= 1e8): the transient state table is built too
            del ts_2
            t0 = time.perf_counter()
            ts_2 = model2.trajset(trajs2, expect=10 ** 9)
            _lib.logl_st(model2.handle(), ts_2, ss2[:1000], th2[:1000], tid2[:1000])
            build_s = time.perf_counter() - t0
            per, kms_, wms_, fr_, _ = time_resident(model2.handle(), ts_2, d1, d2, d3, n_traj2 * per2, k + 1, 10)
            b2, ms2 = _lib.prefix_info(ts_2)
            result['config2_one_gpu']['declared_for_1e9_evaluations'] = {
                'value': n_traj2 * per2 / per, 'unit': 'evals/s', 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_,
                'tables_bytes': b2, 'tables_build_ms_device': ms2, 'upload_and_first_evaluation_s': build_s}
            del d1, d2, d3, ts_2, model2
        except Exception as exc:   # (a side measurement must not take the headline down with it)
            result['config2_one_gpu'] = {'error': repr(exc)}

        # ---- BASELINE configs[3]: 3-state, T = 2000, 5000 candidates per trajectory, mixed missing-frame masks ----------
        try:
            rng3 = np.random.default_rng(33)
            model3 = bild_amd.MultiStateRouse(20, 1., 5., d=3, looppositions=H.LOOPS[3], localization_error=0.1)
            T3, per3, kinds = 2000, 5000, ['none', 'iid', 'bursty', 'none', 'iid', 'bursty']
            trajs3 = []
            for j, kind in enumerate(kinds):
                miss = H.missing_mask(rng3, T3, kind)
                if j % 2 == 1:
                    miss = np.union1d(miss, [0])                      # frame 0 missing in half of them
                trajs3.append(model3.trajectory_from_loopingprofile(H.random_profile(rng3, T3, 3, T3 // 5), missing_frames=miss, rng=rng3))
            tid3 = np.repeat(np.arange(len(kinds)), per3).astype(np.int32)
            t0 = time.perf_counter()
            ts3 = model3.trajset(trajs3)
            ss3, th3 = H.candidate_profiles(rng3, 500, 4, 3)
            _lib.logl_st(model3.handle(), ts3, ss3, th3, tid3[:500])
            first3 = time.perf_counter() - t0
            b3, ms3 = _lib.prefix_info(ts3)
            c3 = {'tables_bytes': b3, 'tables_build_ms_device': ms3, 'upload_and_first_evaluation_s': first3}
            for kk in (4, 8):
                ss3, th3 = H.candidate_profiles(rng3, len(kinds) * per3, kk, 3)
                _lib.logl_st(model3.handle(), ts3, ss3[:500], th3[:500], tid3[:500])
                d1, d2, d3 = resident(ss3, th3, tid3)
                per, kms_, wms_, fr_, out3 = time_resident(model3.handle(), ts3, d1, d2, d3, len(tid3), kk + 1, 10)
                pick = rng3.choice(len(tid3), 6, replace=False)
                from oracle import oracle
                worst = 0.0
                for r_ in pick:
                    st_ = H.expand(ss3[r_:r_ + 1], th3[r_:r_ + 1], T3)
                    want = oracle.logl_batch(model3.arrays(), model3.measurement, model3.localization_error, trajs3[tid3[r_]][:], st_)[0]
                    worst = max(worst, abs(float(out3[r_].item()) - want))
                c3[f'k={kk}'] = {'evals_per_s': len(tid3) / per, 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_, 'walk_kernel_ms': wms_,
                                 'frames_executed_fraction': fr_ / (len(tid3) * T3), 'max_abs_diff_vs_oracle_6_candidates': worst}
            result['config3'] = {'what': f'BASELINE configs[3]: 3-state Rouse, T={T3}, {per3} candidates per trajectory on {len(kinds)} trajectories '
                                         '(no missing frames / 10 % i.i.d. / bursty gaps covering 30 %, frame 0 missing in half), resident in HBM', **c3}
            del ts3, model3
        except Exception as exc:
            result['config3'] = {'error': repr(exc)}

        # ---- BASELINE configs[4]: the full adaptive-k inference on 64 trajectories of experimental length ---------------
        try:
            rng4 = np.random.default_rng(5)
            model4 = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
            trajs4 = [model4.trajectory_from_loopingprofile(H.random_profile(rng4, int(rng4.integers(150, 601)), 2, 120), rng=rng4)
                      for _ in range(64)]
            model4.logL_segments(np.zeros((1, 1), np.int32), np.zeros((1, 1), np.int32), trajs4, np.zeros(1, np.int32))   # upload

            def run4(**kw4):
                _lib.kernel_timing(True)
                t0 = time.perf_counter()
                res = bild_amd.sample_many(trajs4, model4, return_exceptions=True, **kw4)
                wall = time.perf_counter() - t0
                _lib.kernel_timing(False)
                kms, launches, _ = _lib.kernel_timing_read()
                wms, wl = _lib.kernel_timing_read_walk()
                ok = [r for r in res if not isinstance(r, Exception)]
                evals = sum(len(smp['logLs']) for r in ok for s_ in r.samplers for smp in s_.samples)
                return {'wall_s': wall, 'trajectories_done': len(ok), 'amis_steps': sum(len(r.log['k']) for r in ok),
                        'likelihood_evaluations': evals, 'evals_per_s': evals / wall, 'gpu_busy_ms': kms + wms,
                        'gpu_busy_share': (kms + wms) * 1e-3 / wall, 'kernel_launches': launches + wl,
                        'best_k_histogram': np.bincount([int(r.best_k()) for r in ok]).tolist()}
            np.random.seed(11)
            run4()                                  # (the per-k constants of the transition matrix, cached per process: 10 ms)
            np.random.seed(11)
            c4 = run4()
            c4_gen = run4(rng=np.random.default_rng(11))
            np.random.seed(11)
            c4_py = run4(driver='python')
            result['config4'] = {
                'what': 'BASELINE configs[4]: bild.core.sample with default settings (adaptive k, N = 100 per AMIS step) on 64 synthetic '
                        'trajectories of T ~ U{150..600} through sample_many: the native inference driver (csrc/run_host.cpp) -- one '
                        'round = one likelihood call over the pending candidates of ALL trajectories + their bookkeeping on host '
                        'threads; random numbers from the global NumPy stream (the reference\'s), three bulk draws per round; one GPU',
                **c4,
                'with_numpy_generator': dict(c4_gen, what='the same with rng=np.random.default_rng(11): the three bulk draws per round through '
                                                          'a numpy Generator instead of the legacy global stream'),
                'python_driver': dict(c4_py, what="driver='python': round 3's way -- one Python loop per trajectory as cooperative "
                                                  "tasks, pending batches fused into one launch")}
            del model4
        except Exception as exc:
            result['config4'] = {'error': repr(exc)}

    if secondary and args.sweep:
        # ---- chain lengths x number of distinct localization errors, at HEAD ---------------------------------
        rows = {}
        for Nm in (4, 8, 16, 20, 32):
            for errs, name in ((0.1, 'd*=1'), ([0.1, 0.1, 0.25], 'd*=2')):
                rng_s = np.random.default_rng(Nm)
                mod = bild_amd.MultiStateRouse(Nm, 1., 5., d=3, localization_error=errs)
                tr = mod.trajectory_from_loopingprofile(H.random_profile(rng_s, T, 2, T // 5), rng=rng_s)
                ss_s, th_s = H.candidate_profiles(rng_s, n, k, 2)
                hs, tss = mod.handle(), mod.trajset(tr)
                _lib.logl_st(hs, tss, ss_s[:200], th_s[:200])
                d1, d2, _ = resident(ss_s, th_s)
                per, kms_, wms_, fr_, _ = time_resident(hs, tss, d1, d2, None, n, k + 1, 20)
                can_s, _ = _lib.flop_count(hs, tss, n)
                rows[f'N={Nm} {name}'] = {'evals_per_s': n / per, 'ms_per_step': per * 1e3, 'frame_loop_kernel_ms': kms_, 'walk_kernel_ms': wms_,
                                          'canonical_equiv_frac': can_s / per / 1e12 / FP64_PEAK_TFLOPS, 'modes': hs.query(_lib.Q_NEFF),
                                          'frames_run_per_candidate': fr_ / n, 'tables_bytes': _lib.prefix_info(tss)[0]}
        result['sweep'] = {'what': f'{n} candidates x T={T}, k={k}: chain length N and distinct localization errors (SURVEY 8d)', **rows}

    if rank == 0 and world == 1 and args.scaling == 'weak':
        def seg_step(path, **kw):
            # the segment entry on host-converted lists of batch 0 (frame-by-frame / dense / canonical comparisons)
            _lib.logl_segments_device(seg_h[0], seg_h[1], n, k + 1, seg_d[0].data_ptr(), seg_d[1].data_ptr(), 0, d_out.data_ptr(),
                                      stream=stream_of(), path=path, **kw)
        from bild_amd.profiles import segments_from_st
        a0, b0 = segments_from_st(ss, thetas, T)
        seg_d = (torch.from_numpy(a0).to(dev), torch.from_numpy(b0).to(dev))
        seg_h = [h, ts]

        def default_results():
            turn[0] = 0
            step()
            torch.cuda.synchronize()
            return d_out[:n].cpu().numpy().copy()
        if not args.no_secondary and prefix_bytes:
            reps = max(5, args.steps // 3)
            ndt, nkms, nname = timed(lambda: seg_step(args.path, prefix=False), reps, 1)
            _, nexe = _lib.flop_count(h, ts, n, traj_id=traj_id, path=args.path)
            a_out = d_out[:n].cpu().numpy().copy()
            result['without_tables'] = {
                'what': 'batch 0 with every candidate run frame by frame from frame 0 (BILD_NO_PREFIX, array order): the frame loop of '
                        'the same kernel code in its issue-bound regime',
                'value': n * reps / ndt, 'unit': 'evals/s', 'kernel': nname, 'kernel_ms': nkms,
                'achieved': nexe / (nkms * 1e-3) / 1e12, 'frac': nexe / (nkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                'max_abs_diff_vs_default': float(np.max(np.abs(a_out - default_results())))}
            roofline['same_kernel_frame_by_frame'] = {'kernel_ms': nkms, 'achieved': nexe / (nkms * 1e-3) / 1e12,
                                                      'frac': nexe / (nkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                                      'note': 'every frame of every candidate run (no tables): what the frame loop itself '
                                                              'reaches of the fp64 peak'}
            sdt_, skms_, _ = timed(lambda: seg_step(args.path, split=False, states=False), reps, 1)
            result['single_launch'] = {
                'what': 'batch 0 as ONE launch of the frame-loop kernel (BILD_NO_SPLIT | BILD_NO_STATES: round 2\'s default) on '
                        'host-converted segment lists resident in HBM',
                'value': n * reps / sdt_, 'unit': 'evals/s', 'kernel_ms': skms_,
                'max_abs_diff_vs_default': float(np.max(np.abs(d_out[:n].cpu().numpy() - default_results())))}
        if not args.no_secondary and args.path != 'dense':
            reps = max(3, args.steps // 10)
            ddt, dkms, dname = timed(lambda: seg_step('dense'), reps, 1)
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
            seg_h[0], seg_h[1] = h_full, ts_full
            cdt, ckms, cname = timed(lambda: seg_step('dense'), 3, 1)
            ccan, cexe = _lib.flop_count(h_full, ts_full, n, path='dense')
            full_out = d_out[:n].cpu().numpy().copy()
            seg_h[0], seg_h[1] = h, ts
            result['canonical_path'] = {
                'what': 'dense path without reduction: the reference recursion itself on all %d monomers' % h_full.query(_lib.Q_N),
                'value': n * 3 / cdt, 'unit': 'evals/s', 'kernel': cname, 'kernel_ms': ckms, 'bound': 'mfma' if 'mfma' in cname else 'valu',
                'achieved': cexe / (ckms * 1e-3) / 1e12, 'unit_achieved': 'TFLOP/s fp64',
                'frac': cexe / (ckms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                'max_abs_diff_vs_default_path': float(np.max(np.abs(full_out - default_results())))}
        if not args.no_cpu_baseline:
            base, ref_out = cpu_baseline(rank, n, T, k, args.states)
            result['cpu_baseline'] = base
            got = default_results()[:len(ref_out)]
            result['parity_max_abs_diff_vs_cpu_baseline'] = float(np.max(np.abs(got - ref_out)))
            result['speedup_vs_cpu_baseline'] = value / base['value']
            if base.get('reference_equiv'):
                result['speedup_vs_reference_equiv'] = value / base['reference_equiv']['value']
            if args.cpu_allcores:
                cores = _cpu_share()
                allc, _ = cpu_baseline(rank, n, T, k, args.states, budget_s=10.0, procs=cores)
                result['cpu_baseline_allcores'] = allc
                result['speedup_vs_cpu_allcores'] = value / allc['value']

    if direct is not None:
        torch.cuda.synchronize()
        direct.status()         # a peer that did not deliver within the timeout is an error, not a number
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
