#!/usr/bin/env python3
"""
Headline benchmark: logL evaluations / second of one AMIS batch on MI355X.

Workload (BASELINE.json configs[1]): a batch of 10 000 candidate looping profiles x 1
trajectory, T = 1000 frames, 2-state Rouse model, N = 20 monomers, d = 3, one localization
error (d* = 1), k = 4 switches per candidate, fp64.  One "step" = one pass of the hot path
over one such batch: everything `FixedkSampler.logL(ss, thetas)` does for one AMIS iteration
(reference bild/amis.py:717-739), with the run-length encoded profiles and the trajectory
already resident in HBM.  With N > 1 GPUs every rank evaluates its own 10k batch (weak
scaling, one process per GPU) and the per-step result is exchanged with ONE all-gather of
the log-likelihoods (RCCL), as an AMIS step needs them to form the importance weights.

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


def build_workload(rank, n_samples, T, k, S=2, N=20, d=3, err=0.1):
    import helpers as H
    import bild_amd
    rng_t = np.random.default_rng(1000 + rank)
    rng_c = np.random.default_rng(2000 + rank)
    model = bild_amd.MultiStateRouse(N, 1., 5., d=d, looppositions=H.LOOPS[S], localization_error=err)
    truth = H.random_profile(rng_t, T, S, T // 5)
    traj = model.trajectory_from_loopingprofile(truth, rng=rng_t)
    ss, thetas = H.candidate_profiles(rng_c, n_samples, k, S)
    return model, traj, ss, thetas


def _cpu_baseline_loop(model, traj, ss, thetas, T, budget_s):
    """ the timed loop itself; runs in a child process (see cpu_baseline) """
    import helpers as H
    from oracle import oracle
    ref = oracle.load_reference_cython()
    kind = 'reference'
    states = H.expand(ss[:8192], thetas[:8192], T)

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
    """ `bench.py --cpu-baseline-child rank n T k S budget out.npz`: CPU only, never touches the GPU """
    rank, n, T, k, S = (int(v) for v in argv[:5])
    model, traj, ss, thetas = build_workload(rank, n, T, k, S=S)
    kind, dt, out = _cpu_baseline_loop(model, traj, ss, thetas, T, float(argv[5]))
    np.savez(argv[6], kind=kind, dt=dt, out=out)


def cpu_baseline(rank, n, T, k, S, budget_s=15.0):
    """
    The reference's own Cython kernel (compiled unmodified into oracle/_ref) on ONE host core,
    driven exactly like FixedkSampler.logL drives it (a Python loop, amis.py:735-739), on a
    bounded sample of the same batch.  It runs in a fresh child process with the BLAS pools pinned to
    one thread from the start: inside this process (torch loaded, pools limited after the fact) the
    same loop is 20-35 % slower, which would flatter the GPU.
    """
    import subprocess
    import tempfile
    env = dict(os.environ, OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'cpu_baseline.npz')
        subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child',
                        str(rank), str(n), str(T), str(k), str(S), str(budget_s), path],
                       env=env, check=True, timeout=budget_s + 600)
        with np.load(path) as z:
            kind, dt, out = str(z['kind']), float(z['dt']), np.array(z['out'])
    return dict(value=len(out) / dt, unit='evals/s', cores=1, kind=kind,
                sample=f"first {len(out)} profiles of the rank-0 batch, T={T}, {dt:.1f} s of "
                       f"{'reference Cython MSRouse_logL (oracle/_ref)' if kind == 'reference' else 'oracle C port'}"
                       f" in a Python loop (amis.py:735-739), own process, BLAS pinned to 1 thread"), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--samples', type=int, default=10000, help='candidate profiles per GPU per step')
    ap.add_argument('--T', type=int, default=1000)
    ap.add_argument('--k', type=int, default=4)
    ap.add_argument('--states', type=int, default=2)
    ap.add_argument('--path', default='auto', choices=['auto', 'modal', 'dense'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-reduce', action='store_true', help='keep all N modes (skip the invariant-subspace reduction)')
    ap.add_argument('--no-dense', action='store_true', help='skip the secondary dense-path measurement')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="collective backend; 'gloo' (log-likelihoods staged through host memory) rehearses the "
                         "multi-rank path on a box with fewer GPUs than ranks")
    ap.add_argument('--host-buffers', action='store_true',
                    help='also time the host-buffer entry point (H2D of the profiles + D2H of the results per step)')
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

    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    from bild_amd import dist as bdist

    n, T, k = args.samples, args.T, args.k
    model, traj, ss, thetas = build_workload(rank, n, T, k, S=args.states)
    model.path = args.path
    if args.no_reduce:
        a_ = model.arrays()
        model._handle = _lib.ModelHandle(a_['B'], a_['G'], a_['Sig'], a_['M0'], a_['C0'], model.measurement, reduce=False)
    h = model.handle()
    ts = model.trajset(traj)                          # trajectory resident in HBM
    seg_start, seg_state = segments_from_st(ss, thetas, T)
    dev = torch.device('cuda', dev_index)
    d_start = torch.from_numpy(seg_start).to(dev)     # candidates resident in HBM
    d_state = torch.from_numpy(seg_state).to(dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    d_all = torch.empty(n * world, dtype=torch.float64, device=dev)

    def step(path):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.logl_segments_device(h, ts, n, k + 1, d_start.data_ptr(), d_state.data_ptr(), 0, d_out.data_ptr(),
                                  stream=stream, path=path)
        if world > 1:                                 # the one collective of an AMIS step
            if args.backend == 'nccl':
                bdist.all_gather_logl(d_out, d_all)
            else:
                d_all.copy_(bdist.all_gather_logl(d_out.cpu()))

    def timed(path, steps, warmup):
        for _ in range(warmup):
            step(path)
        _lib.kernel_timing(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(path)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        _lib.kernel_timing(False)
        kms, launches, kname = _lib.kernel_timing_read()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms / max(launches, 1), kname

    dt, kernel_ms, kname = timed(args.path, args.steps, args.warmup)
    value = world * n * args.steps / dt

    # HBM bytes per launch measured with rocprofv3 PMC passes (profiles/*_hbm_traffic.json), quoted only
    # when the committed measurement is for this very workload
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')) as f:
            tj = json.load(f)
        if tj['workload'] == {'samples': n, 'T': T, 'k': k, 'path': args.path} and args.states == 2:
            traffic = tj['traffic_bytes_raw']
    except Exception:
        pass
    can, exe = _lib.flop_count(h, ts, n, path=args.path)
    ksec = kernel_ms * 1e-3
    roofline = {
        'bound': 'mfma',
        'pipe': 'fp64 vector FMA (v_fma_f64); the gfx950 f64 matrix peak is the same 78.6 TFLOP/s',
        'kernel': kname,
        'achieved': can / ksec / 1e12, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': can / ksec / 1e12 / FP64_PEAK_TFLOPS,
        'traffic': traffic,
        'traffic_unit': 'bytes per launch (FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, profiles/r01_hbm_traffic.json)',
        'flops_basis': 'canonical F of SURVEY 8a (dense recursion on all N monomers) x evaluations per launch',
        'flop_per_eval_canonical': can / n,
        'flop_per_eval_executed': exe / n,
        'executed_achieved': exe / ksec / 1e12,
        'executed_frac': exe / ksec / 1e12 / FP64_PEAK_TFLOPS,
        'kernel_ms': kernel_ms,
        'attainable_fma_peak': 51.5,
        'attainable_note': 'register-resident v_fma_f64 loop measured on MI355X: 51.5 TFLOP/s at >= 2 waves/SIMD, 34.1 at one (profiles/r01_f64_rates.txt)',
        'hbm_algorithmic_GBps': (n * (k + 1) * 8 + n * 8 + T * 3 * 8) / ksec / 1e9,
        'hbm_frac': (n * (k + 1) * 8 + n * 8 + T * 3 * 8) / ksec / 1e9 / HBM_PEAK_GBS,
        'note': ('path=%s runs the recursion in %d of %d modes (invariant-subspace reduction) and, on the modal '
                 'path, in the eigenbasis of B (elementwise predict): frac > 1 means fewer operations were '
                 'executed than the canonical count, executed_frac is the fraction of the fp64 peak the '
                 'instructions actually issued reach') % (args.path, h.query(_lib.Q_NEFF), h.query(_lib.Q_N)),
    }

    result = {
        'metric': 'logL evaluations/sec (T=1000, 2-state) per AMIS batch',
        'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'configs[1]: {n} profile samples x 1 trajectory per GPU, T={T}, {args.states}-state '
                               f'Rouse N=20 d=3 d*=1, k={k} switches, fp64',
                   'samples_per_gpu': n, 'T': T, 'k': k, 'states': args.states, 'path': args.path,
                   'collective': ('all_gather(float64[%d]) per step, %s' % (n, args.backend)) if world > 1 else 'none (1 GPU)'},
        'roofline': roofline,
    }

    if rank == 0 and world == 1:
        if not args.no_dense and args.path != 'dense':
            ddt, dkms, dname = timed('dense', max(3, args.steps // 10), 1)
            dcan, dexe = _lib.flop_count(h, ts, n, path='dense')
            result['dense_path'] = {
                'value': n * max(3, args.steps // 10) / ddt, 'unit': 'evals/s', 'kernel': dname, 'kernel_ms': dkms,
                'achieved': dcan / (dkms * 1e-3) / 1e12, 'frac': dcan / (dkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                'executed_frac': dexe / (dkms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            }
        if not args.no_dense and not args.no_reduce:
            # the reference's literal algorithm: all N monomers, C <- B C B + Sig every frame
            # (canonical flop count == executed flop count)
            a_ = model.arrays()
            h_full = _lib.ModelHandle(a_['B'], a_['G'], a_['Sig'], a_['M0'], a_['C0'], model.measurement, reduce=False)
            ts_full = _lib.TrajSetHandle(h_full, [np.asarray(traj[:])], np.asarray(model.localization_error)[None, :])
            reps = 3
            _lib.kernel_timing(False)
            for _ in range(1):
                _lib.logl_segments_device(h_full, ts_full, n, k + 1, d_start.data_ptr(), d_state.data_ptr(), 0,
                                          d_out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, path='dense')
            _lib.kernel_timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                _lib.logl_segments_device(h_full, ts_full, n, k + 1, d_start.data_ptr(), d_state.data_ptr(), 0,
                                          d_out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, path='dense')
            torch.cuda.synchronize()
            cdt = time.perf_counter() - t0
            _lib.kernel_timing(False)
            ckms, claunches, cname = _lib.kernel_timing_read()
            ckms /= max(claunches, 1)
            ccan, cexe = _lib.flop_count(h_full, ts_full, n, path='dense')
            full_out = d_out.cpu().numpy().copy()
            result['canonical_path'] = {
                'what': 'dense path without reduction: the reference recursion itself on all %d monomers' % h_full.query(_lib.Q_N),
                'value': n * reps / cdt, 'unit': 'evals/s', 'kernel': cname, 'kernel_ms': ckms,
                'achieved': ccan / (ckms * 1e-3) / 1e12, 'unit_achieved': 'TFLOP/s fp64',
                'frac': ccan / (ckms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            }
            step(args.path)
            torch.cuda.synchronize()
            result['canonical_path']['max_abs_diff_vs_default_path'] = float(np.max(np.abs(full_out - d_out.cpu().numpy())))
        if args.host_buffers:
            # PCIe-inclusive rate of the synchronous host entry point (never `value`)
            for _ in range(2):
                _lib.logl_segments(h, ts, seg_start, seg_state, path=args.path)
            t0 = time.perf_counter()
            reps = max(5, args.steps // 2)
            for _ in range(reps):
                _lib.logl_segments(h, ts, seg_start, seg_state, path=args.path)
            result['host_buffers'] = {'value': n * reps / (time.perf_counter() - t0), 'unit': 'evals/s',
                                      'note': 'bild_logl_segments: H2D profiles + launch + D2H results + sync per step'}
        if not args.no_cpu_baseline:
            base, ref_out = cpu_baseline(rank, n, T, k, args.states)
            result['cpu_baseline'] = base
            step(args.path)
            torch.cuda.synchronize()
            got = d_out[:len(ref_out)].cpu().numpy()
            result['parity_max_abs_diff_vs_cpu_baseline'] = float(np.max(np.abs(got - ref_out)))
            result['speedup_vs_cpu_baseline'] = value / base['value']

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
