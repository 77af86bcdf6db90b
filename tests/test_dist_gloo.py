"""
World-size-2 tests of the multi-GPU sharding logic on CPU (gloo): contiguous shards cover the
batch exactly once and one all-gather per step reassembles the full log-likelihood vector in
sample order.  The per-shard evaluator here is a deterministic stand-in (no GPU in this tier);
the real kernel is exercised by the -m gpu tests.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_logl(idx):
    return np.sin(idx.astype(np.float64)) * 1e3 - idx


def _worker(rank, world, port, n, ragged, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from bild_amd import dist as bdist
    lo, hi = bdist.shard_bounds(n, world, rank)
    local = torch.from_numpy(_fake_logl(np.arange(lo, hi)))
    if ragged:
        sizes = [bdist.shard_bounds(n, world, r)[1] - bdist.shard_bounds(n, world, r)[0] for r in range(world)]
        full = bdist.all_gather_logl_ragged(local, sizes)
    else:
        full = bdist.all_gather_logl(local)
    ok = np.array_equal(full.numpy(), _fake_logl(np.arange(n)))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n,ragged', [(10000, False), (10001, True), (7, True)])
def test_shard_and_allgather_world2(n, ragged):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), n, ragged, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_bounds_cover_exactly_once():
    from bild_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 9, 10000, 256000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_trajectory_balances_cost():
    from bild_amd.dist import shard_by_trajectory
    rng = np.random.default_rng(0)
    T = rng.integers(150, 600, size=64)
    n = np.full(64, 1000)
    owners = shard_by_trajectory(T, n, 8)
    assert sorted(np.concatenate(owners).tolist()) == list(range(64))
    load = np.array([np.sum(T[o] * n[o]) for o in owners], dtype=float)
    assert load.max() / load.mean() < 1.05


def _amis_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bild_amd
    from bild_amd.dist import ShardedModel
    from amis_cases import _table
    from test_core import _SegmentTableModel
    table = _table(7, 2, 40, [11, 29])
    inner = _SegmentTableModel([table])
    traj = bild_amd.Trajectory(np.zeros((40, 1)))
    np.random.seed(123)                     # replicated loop: same seed on every rank
    sampler = bild_amd.FixedkSampler(traj, ShardedModel(inner), k=2, N=37, max_fcomplete=10)
    for _ in range(4):
        sampler.step()
    ret[rank] = (np.array(sampler.evidences), np.concatenate([s['logLs'] for s in sampler.samples]))
    dist.barrier()
    dist.destroy_process_group()


def test_replicated_amis_with_sharded_likelihood_world2():
    """ the AMIS loop replicated on 2 ranks, each evaluating half of every batch + one all-gather per step """
    import bild_amd
    from amis_cases import _table
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_core import _SegmentTableModel
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_amis_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        (ev0, l0), (ev1, l1) = ret[0], ret[1]
    assert np.array_equal(ev0, ev1) and np.array_equal(l0, l1)          # ranks stay in lockstep
    # ... and equal to the single-process run
    table = _table(7, 2, 40, [11, 29])
    traj = bild_amd.Trajectory(np.zeros((40, 1)))
    np.random.seed(123)
    ref = bild_amd.FixedkSampler(traj, _SegmentTableModel([table]), k=2, N=37, max_fcomplete=10)
    for _ in range(4):
        ref.step()
    assert np.array_equal(np.array(ref.evidences), ev0)
    assert np.array_equal(np.concatenate([s['logLs'] for s in ref.samples]), l0)


def _routed_model(tables):
    from test_core import _SegmentTableModel

    class Routed(_SegmentTableModel):      # the fused evaluator sees local trajectory ids: route by the trajectory itself
        def logL_segments(self, seg_start, seg_state, trajs_, traj_id):
            ids = np.array([int(trajs_[j][0, 0]) for j in traj_id])
            return super().logL_segments(seg_start, seg_state, None, ids)

        def __reduce__(self):              # results carry their model: make this test double picklable
            return (_routed_model, (self.tables,))
    return Routed(tables)


def _many_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bild_amd
    from bild_amd.dist import sample_many_distributed
    from amis_cases import _table
    from test_core import _SegmentTableModel
    tables = [_table(50 + j, 2, 14 + 5 * j, [4 + j, 9 + 2 * j]) for j in range(5)]
    trajs = [bild_amd.Trajectory(np.full((t.shape[1], 1), float(j))) for j, t in enumerate(tables)]

    res = sample_many_distributed(trajs, _routed_model(tables), seed=5, init_runs=2, k_max=3,
                                  sampler_kw={'N': 20, 'max_fev': 200, 'max_fcomplete': 30}, choice_kw={'samplesize': 300})
    ret[rank] = [(len(r.traj), r.best_k(), np.array(r.evidence), r.best_profile()[:]) for r in res]
    dist.barrier()
    dist.destroy_process_group()


def test_sample_many_distributed_world2():
    """ trajectories sharded over 2 ranks, one object all-gather at the end: every rank holds every result, in order """
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_many_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        r0, r1 = ret[0], ret[1]
    assert [x[0] for x in r0] == [14, 19, 24, 29, 34]                    # all five results, in the order of the input
    for a, b in zip(r0, r1):
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2], equal_nan=True) and np.array_equal(a[3], b[3])
        assert len(a[3]) == a[0] and np.all(np.isfinite(a[2][:2]))


_RCCL_SCRIPT = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import helpers as H, bild_amd
from bild_amd import _lib, dist as bdist
from bild_amd.profiles import segments_from_st
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.default_rng(3)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 300, 2, 60), rng=rng)
ss, thetas = H.candidate_profiles(rng, 257, 3, 2)
a, b = segments_from_st(ss, thetas, 300)
dev = torch.device("cuda", 0)
d_a, d_b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
d_out = torch.empty(257, dtype=torch.float64, device=dev)
_lib.logl_segments_device(model.handle(), model.trajset(traj), 257, 4, d_a.data_ptr(), d_b.data_ptr(), 0,
                          d_out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
full = bdist.all_gather_logl(d_out)                       # device tensors through RCCL
ragged = bdist.all_gather_logl_ragged(d_out, [257])
host = model.logL_st_batch(ss, thetas, traj)
# the product path of a multi-GPU AMIS step: kernel -> device buffer -> all-gather of that buffer -> ONE host copy
sm = bdist.ShardedModel(model, collective_at_world1=True)
sharded = sm.logL_st_batch(ss, thetas, traj)
sharded2 = sm.logL_st_batch(ss[:100], thetas[:100], traj)
np.random.seed(4)
sampler = bild_amd.FixedkSampler(traj, sm, k=3, N=64, max_fcomplete=0)
before = sm.host_copies
for _ in range(3):
    sampler.step()
steps_ok = sm.host_copies - before == 3
np.random.seed(4)
plain = bild_amd.FixedkSampler(traj, model, k=3, N=64, max_fcomplete=0)
for _ in range(3):
    plain.step()
dist.barrier()
ok = (np.array_equal(full.cpu().numpy(), host) and np.array_equal(ragged.cpu().numpy(), host)
      and np.array_equal(sharded, host) and np.array_equal(sharded2, host[:100]) and np.all(np.isfinite(host))
      and steps_ok and np.array_equal(np.array(sampler.evidences), np.array(plain.evidences)))
dist.destroy_process_group()
print("RCCL_OK" if ok else "RCCL_MISMATCH")
'''


def test_sharded_model_pickles_and_copies():
    """ the wrapper's attribute forwarding must not recurse while an instance is being rebuilt (no GPU needed) """
    import copy
    import pickle
    from bild_amd.dist import ShardedModel
    from test_core import _SegmentTableModel
    from amis_cases import _table
    import bild_amd
    inner = _SegmentTableModel([_table(7, 2, 40, [11, 29])])
    sm = ShardedModel(inner)
    for clone in (copy.deepcopy(sm), pickle.loads(pickle.dumps(sm)), copy.copy(sm)):
        assert clone.nStates == inner.nStates and clone._group is None
        assert np.array_equal(clone.transitions, inner.transitions)
    with pytest.raises(AttributeError):
        sm.no_such_attribute
    # a sampler (and its results) holding the wrapper as .model survives the same
    np.random.seed(1)
    sampler = bild_amd.FixedkSampler(bild_amd.Trajectory(np.zeros((40, 1))), sm, k=1, N=10, max_fcomplete=0)
    sampler.step()
    back = pickle.loads(pickle.dumps(sampler))
    assert np.array_equal(np.array(back.evidences), np.array(sampler.evidences))
    assert isinstance(copy.deepcopy(sampler).model, ShardedModel)


@pytest.mark.gpu
def test_rccl_allgather_of_device_results(built_lib):
    """
    The collective of the multi-GPU path on its real backend (nccl = RCCL), world size 1 -- the one-GPU
    box cannot hold more ranks on RCCL: device-resident kernel results go through `all_gather_logl`
    unchanged.  Runs in a child process so the process group does not outlive the test.
    """
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_LIBCOMM_SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import helpers as H, bild_amd
from bild_amd import _lib, dist as bdist
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 300, 2, 60), rng=rng)
ss, thetas = H.candidate_profiles(rng, 321, 3, 2)
host = model.logL_st_batch(ss, thetas, traj)
comm = bdist.LibraryComm.from_file(sys.argv[2], 1, 0, nonce=str(os.getpid()))            # rank 0 of 1: writes the id file and reads it back
sm = bdist.ShardedModel(model, comm=comm, collective_at_world1=True)
a = sm.logL_st_batch(ss, thetas, traj)
b = sm.logL_st_batch(ss[:50], thetas[:50], traj)
np.random.seed(4)
sampler = bild_amd.FixedkSampler(traj, sm, k=3, N=64, max_fcomplete=0)
before = sm.host_copies
for _ in range(3):
    sampler.step()
np.random.seed(4)
plain = bild_amd.FixedkSampler(traj, model, k=3, N=64, max_fcomplete=0)
for _ in range(3):
    plain.step()
ok = (np.array_equal(a, host) and np.array_equal(b, host[:50]) and sm.host_copies - before == 3
      and np.array_equal(np.array(sampler.evidences), np.array(plain.evidences)) and "torch" not in sys.modules)
print("LIBCOMM_OK" if ok else "LIBCOMM_MISMATCH", "torch" in sys.modules)
'''


@pytest.mark.gpu
def test_library_collective_without_pytorch(built_lib, tmp_path):
    """
    The multi-GPU step with nothing but the library: shard results stay in HBM (`bild_logl_st_to_device`), the all-gather
    is the library's own RCCL call (`bild_comm_allgather`), one device-to-host copy per step -- and PyTorch is never
    imported.  World size 1 (one GPU on the box); the id travels through a file, as it would between processes.
    """
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _LIBCOMM_SCRIPT, ROOT, str(tmp_path / 'comm.id')], env=env, capture_output=True,
                       text=True, timeout=300)
    assert 'LIBCOMM_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _refused_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bild_amd
    from bild_amd import _lib
    from bild_amd.dist import ShardedModel
    from amis_cases import _table
    from test_core import _SegmentTableModel

    class Strict(_SegmentTableModel):      # refuses what the GPU entry refuses (the conversion raises on the owner's shard)
        def logL_st_batch(self, ss, thetas, traj):
            _lib.segments_from_st(ss, thetas, len(traj), 2)
            return super().logL_st_batch(ss, thetas, traj)
    table = _table(7, 2, 40, [11, 29])
    model = ShardedModel(Strict([table]))
    traj = bild_amd.Trajectory(np.zeros((40, 1)))
    rng = np.random.default_rng(5)
    ss = rng.dirichlet(np.ones(3), size=20)
    thetas = np.tile([0, 1, 0], (20, 1))
    good = model.logL_st_batch(ss, thetas, traj)
    bad = ss.copy()
    bad[15, 0] = -0.2                       # a row of rank 1's shard that is no point on the simplex
    try:
        model.logL_st_batch(bad, thetas, traj)
        outcome = 'no error'
    except _lib.BildAmdError as err:
        outcome = 'refused: ' + str(err)
    again = model.logL_st_batch(ss, thetas, traj)     # the ranks are still in step: the next collective works
    ret[rank] = (outcome, bool(np.array_equal(good, again)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_refused_row_fails_the_step_on_every_rank_world2():
    """ round-3 advisor finding: only the rank that owned a refused row raised; the others went on and hung """
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_refused_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        out = dict(ret)
    assert out[0][0].startswith('refused') and out[0][0] == out[1][0] and 'sample 15' in out[0][0]
    assert out[0][1] and out[1][1]
