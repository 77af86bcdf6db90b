"""
Multi-GPU sharding of an AMIS batch: one process per GPU (``torch.distributed``; backend
"nccl" is RCCL on ROCm, "gloo" on CPU for tests).

The evaluations of a batch are independent (reference bild/amis.py:735-739 loops over them
in arbitrary order), so the batch is split into contiguous shards with no data-path
communication.  The one exchange an AMIS step needs is the vector of log-likelihoods itself
-- every rank forms the importance weights ``logLs - log_delta + log(n_steps)``
(bild/amis.py:843-845) from the full vector -- i.e. a single all-gather of float64[n_local]
per step.  Messages are tiny (80 KB per 10k batch): the collective is latency-bound, so it
is issued once per step on the full shard, never per sample or per trajectory.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """ contiguous, balanced split of range(n): sizes differ by at most one """
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_trajectory(T_per_traj, n_per_traj, world):
    """
    Assign whole trajectories to ranks (keeps one trajectory's data and missing-frame branch
    pattern on one device), balancing the cost model T * n_samples greedily.

    Returns a list of index arrays, one per rank.
    """
    cost = np.asarray(T_per_traj, dtype=np.float64) * np.asarray(n_per_traj, dtype=np.float64)
    order = np.argsort(-cost, kind='stable')
    load = np.zeros(world)
    owner = [[] for _ in range(world)]
    for j in order:
        r = int(np.argmin(load))
        owner[r].append(int(j))
        load[r] += cost[j]
    return [np.array(sorted(o), dtype=np.int64) for o in owner]


def all_gather_logl(local, out=None, group=None):
    """
    All-gather equally sized shards of log-likelihoods (torch tensors on the backend's device).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(local.numel() * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def all_gather_logl_ragged(local, sizes, group=None):
    """
    All-gather shards of different length (``sizes[r]`` entries on rank r): padded to the
    longest shard so that a single collective suffices, then compacted.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    m = int(max(sizes))
    pad = torch.zeros(m, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    buf = torch.empty(m * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * m:r * m + int(sizes[r])] for r in range(world)])


class LibraryComm:
    """
    The step's collective through the library's own RCCL communicator (include/bild_amd.h, "several GPUs"): no PyTorch
    in the multi-GPU path.  The 128-byte communicator id has to reach every rank by some channel of the host program;
    two ready-made ones: a file on a shared file system (`from_file`) or an initialised ``torch.distributed`` group of
    any backend (`from_torch`).  One process per GPU; the device must be current before the communicator is created.
    """

    def __init__(self, world, rank, unique_id):
        from . import _lib
        self.world, self.rank = int(world), int(rank)
        self._comm = _lib.CommHandle(unique_id, world, rank)

    @staticmethod
    def new_id():
        from . import _lib
        return _lib.comm_unique_id()

    @classmethod
    def from_file(cls, path, world, rank, nonce, timeout=120.0):
        """
        Rendezvous through a shared file system.  `nonce` (any string without a path separator: a launch id, a job id,
        the rank-0 pid handed out by the launcher) names THIS job: the id is exchanged through ``path + '.' + nonce``, so a
        file left behind by an earlier run under the same `path` is never read.  Rank 0 refuses to start when the file of
        this very nonce already exists (two jobs with one nonce), and removes it once every rank has created its
        communicator (ranks report through ``<file>.<rank>`` markers).
        """
        import os
        import time
        from . import _lib
        nonce = str(nonce)
        if not nonce or os.sep in nonce:
            raise ValueError("nonce must be a non-empty string without a path separator")
        file = f"{path}.{nonce}"
        if rank == 0:
            if os.path.exists(file):
                raise FileExistsError(f"{file} exists: another job uses the same nonce (or a crashed one left it behind)")
            tmp = file + '.tmp'
            with open(tmp, 'wb') as f:
                f.write(_lib.comm_unique_id())
            os.replace(tmp, file)          # appears atomically
        t0 = time.time()
        while not os.path.exists(file):
            if time.time() - t0 > timeout:
                raise TimeoutError(f"communicator id {file} did not appear")
            time.sleep(0.01)
        with open(file, 'rb') as f:
            uid = f.read()
        comm = cls(world, rank, uid)
        # every rank has the id once its communicator exists: tell rank 0, which then removes all traces
        if rank != 0:
            with open(f"{file}.{rank}", 'wb'):
                pass
        else:
            others = [f"{file}.{r}" for r in range(1, world)]
            while not all(os.path.exists(o) for o in others):
                if time.time() - t0 > timeout:
                    break
                time.sleep(0.01)
            for o in others + [file]:
                try:
                    os.remove(o)
                except OSError:
                    pass
        return comm

    @classmethod
    def from_torch(cls, group=None):
        import torch.distributed as dist
        from . import _lib
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, box[0])

    def allgather(self, d_send, d_recv, n_per_rank, stream=0):
        self._comm.allgather(d_send, d_recv, n_per_rank, stream)


class DirectExchange:
    """
    The step's collective as a one-shot peer write (include/bild_amd.h, "the direct exchange"; csrc/exchange_kernel.hip): ONE
    kernel per rank stores the rank's shard into every peer's receive block and waits for theirs -- for the 10-256 KB a rank
    contributes per AMIS step a ring all-gather is latency-bound, this is one hop.  Same interface as `LibraryComm`
    (`ShardedModel(model, comm=DirectExchange.from_torch(slot))`); the 64-byte IPC handles of the receive blocks have to reach
    every rank once, by any channel: an initialised ``torch.distributed`` group (`from_torch`) or files (`from_files`).  One
    process per GPU; the device must be current before the exchange is created.  ``slot``: the longest shard, in doubles.
    """

    def __init__(self, world, rank, slot, handles=None):
        from . import _lib
        self.world, self.rank, self.slot = int(world), int(rank), int(slot)
        self._x = _lib.ExchangeHandle(world, rank, slot)
        if handles is not None:
            self.connect(handles)

    def handle(self):
        return self._x.handle()

    def connect(self, handles):
        handles = list(handles)
        handles[self.rank] = self._x.handle()
        self._x.connect(handles)

    @classmethod
    def from_torch(cls, slot, group=None):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        self = cls(world, rank, slot)
        handles = [None] * world
        dist.all_gather_object(handles, self.handle(), group=group)
        # every rank maps every block; the verdicts are exchanged in a collective ALL ranks reach whatever happened to them (a
        # rank that raised on its own here would leave the others waiting in theirs), and it is also the point behind which
        # every block is mapped everywhere: nobody stores into one before
        problem = None
        try:
            self.connect(handles)
        except Exception as err:
            problem = f"rank {rank}: {err!r}"
        problems = [None] * world
        dist.all_gather_object(problems, problem, group=group)
        if any(p is not None for p in problems):
            raise RuntimeError("direct exchange: mapping the receive blocks failed: " + "; ".join(p for p in problems if p))
        return self

    @classmethod
    def from_files(cls, path, world, rank, slot, nonce, timeout=120.0):
        """ rendezvous through a shared file system: rank r writes ``<path>.<nonce>.<r>``, reads the others', and all wait for all """
        import os
        import time
        self = cls(world, rank, slot)
        base = f"{path}.{nonce}"
        tmp = f"{base}.{rank}.tmp"
        with open(tmp, 'wb') as f:
            f.write(self.handle())
        os.replace(tmp, f"{base}.{rank}")
        handles, t0 = [], time.time()
        for r in range(world):
            while not os.path.exists(f"{base}.{r}"):
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"exchange handle of rank {r} did not appear")
                time.sleep(0.005)
            with open(f"{base}.{r}", 'rb') as f:
                handles.append(f.read())
        self.connect(handles)
        # second phase: nobody stores into a block before everybody has mapped it
        with open(f"{base}.{rank}.mapped", 'wb'):
            pass
        for r in range(world):
            while not os.path.exists(f"{base}.{r}.mapped"):
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {r} did not map the exchange blocks")
                time.sleep(0.005)
        return self

    def allgather(self, d_send, d_recv, n_per_rank, stream=0):
        self._x.allgather(d_send, d_recv, n_per_rank, stream)

    def status(self):
        self._x.status()


class ShardedModel:
    """
    Multi-GPU likelihood for an AMIS loop that runs replicated on every rank (same seed, hence the
    same proposals everywhere): each rank evaluates its contiguous shard of the batch on its own GPU
    and ONE all-gather per AMIS step gives every rank the full log-likelihood vector, from which all
    ranks form identical weights and refit identical proposals (reference bild/amis.py:843-854).

    With the RCCL backend the shard's results never leave HBM before the collective: the kernel writes them into
    a device buffer (`MultiStateRouse.logL_st_batch_to_device`), that buffer is all-gathered on the same stream,
    and the gathered vector is copied to the host ONCE.  With gloo (CPU rehearsal) the shard comes back through the
    host entry and is gathered there.

    Wraps any model offering ``logL_st_batch``; everything else is forwarded.  With
    ``torch.distributed`` not initialised (or world size 1) it is a transparent pass-through.
    The process group is not part of the pickled state (a copy talks to the default group).
    """

    def __init__(self, model, group=None, device=None, collective_at_world1=False, comm=None):
        self._model = model
        self._group = group
        self._comm = comm      # a `LibraryComm`: shards, all-gather and the host copy go through the library alone
        self._collective_at_world1 = collective_at_world1   # tests: run the sharded path on a single rank too
        self._device = device  # where to stage the gathered vector: None -> cuda for nccl, cpu for gloo
        self._buffers = None   # (local, gathered) device tensors, grown on demand
        self.transitions = model.transitions
        self.host_copies = 0   # device -> host copies made by logL_st_batch (tests assert one per step)

    def __getattr__(self, name):
        # only reached for attributes that are not set on the wrapper itself.  While unpickling / copying the
        # instance exists before its __dict__ does: never look up '_model' through here (it would recurse), and
        # leave dunder lookups (__setstate__, __deepcopy__, ...) to the default machinery.
        if name.startswith('__') or '_model' not in self.__dict__:
            raise AttributeError(name)
        return getattr(self._model, name)

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_group'] = None
        state['_buffers'] = None
        state['_comm'] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1, 0
        return dist.get_world_size(self._group), dist.get_rank(self._group)

    @staticmethod
    def _initialised():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    def logL(self, profile, traj):
        return self._model.logL(profile, traj)

    def _logL_st_batch_library(self, ss, thetas, traj):
        from . import _lib
        comm = self._comm
        n = len(thetas)
        bounds = [shard_bounds(n, comm.world, r) for r in range(comm.world)]
        sizes = [b - a for a, b in bounds]
        lo, hi = bounds[comm.rank]
        m = max(sizes)
        if self._buffers is None or self._buffers[0].n < m:
            self._buffers = (_lib.DeviceBuffer(m), _lib.DeviceBuffer(m * comm.world))
        local, gathered = self._buffers
        if hi > lo:
            self._model.logL_st_batch_to_device(ss[lo:hi], thetas[lo:hi], traj, local.ptr, stream=0)
        comm.allgather(local.ptr, gathered.ptr, m, stream=0)          # default stream: ordered behind the kernel
        host = gathered.to_host(m * comm.world)                       # the one device -> host copy of the step
        self.host_copies += 1
        if hasattr(comm, 'status'):
            comm.status()                                             # (direct exchange: a peer that never delivered)
        self._verdict(host, hi > lo, ss, thetas, traj)
        return np.concatenate([host[r * m:r * m + sizes[r]] for r in range(comm.world)])

    def _verdict(self, gathered, had_shard, ss, thetas, traj):
        """
        Rows the device refused (no points on the simplex) must fail the step on EVERY rank, as the unsharded `logL` does --
        were only the rank that owns the row to raise, the others would go on to the next collective and hang there.  A
        refused row comes back as NaN, and every rank holds the full batch (the AMIS loop is replicated): whenever the
        gathered vector contains a NaN, every rank converts the full batch on the host (`bild_segments_from_st`, the
        conversion the kernels mirror), which raises for exactly the rows the device refuses -- the same exception, for
        the same row, everywhere.  A NaN that a legitimate row produced costs one conversion and passes through.
        """
        from . import _lib
        refused = None
        if had_shard:
            try:
                self._model.check_st_rows()      # waits for ALL pending to_device calls of the model and forgets the verdict
            except _lib.BildAmdError as err:
                refused = err
        if np.isnan(gathered).any():
            _lib.segments_from_st(np.asarray(ss, dtype=np.float64), np.asarray(thetas), len(traj), int(self._model.nStates))
        if refused is not None:     # (cannot happen without a NaN in the gathered vector; never swallow a verdict)
            raise refused

    def logL_st_batch(self, ss, thetas, traj):
        if self._comm is not None:
            if self._comm.world == 1 and not self._collective_at_world1:
                return self._model.logL_st_batch(ss, thetas, traj)
            return self._logL_st_batch_library(np.asarray(ss), np.asarray(thetas), traj)
        world, rank = self._world()
        if world == 1 and not (self._collective_at_world1 and self._initialised()):
            return self._model.logL_st_batch(ss, thetas, traj)
        import torch
        import torch.distributed as dist
        n = len(thetas)
        bounds = [shard_bounds(n, world, r) for r in range(world)]
        sizes = [b - a for a, b in bounds]
        lo, hi = bounds[rank]
        device = self._device
        if device is None:
            device = 'cuda' if dist.get_backend(self._group) == 'nccl' else 'cpu'
        if str(device).startswith('cuda') and hasattr(self._model, 'logL_st_batch_to_device'):
            m = max(sizes)
            if self._buffers is None or self._buffers[0].numel() < m:
                self._buffers = (torch.zeros(m, dtype=torch.float64, device=device),
                                 torch.empty(m * world, dtype=torch.float64, device=device))
            local, gathered = self._buffers[0][:m], self._buffers[1][:m * world]
            if hi > lo:
                self._model.logL_st_batch_to_device(ss[lo:hi], thetas[lo:hi], traj, local.data_ptr(),
                                                    stream=torch.cuda.current_stream().cuda_stream)
            dist.all_gather_into_tensor(gathered, local, group=self._group)      # same stream: ordered behind the kernel
            host = gathered.cpu().numpy()                                        # the one device -> host copy of the step
            self.host_copies += 1
            self._verdict(host, hi > lo, ss, thetas, traj)
            return np.concatenate([host[r * m:r * m + sizes[r]] for r in range(world)])
        local, refused = np.empty(0), None
        if hi > lo:
            try:
                local = np.asarray(self._model.logL_st_batch(ss[lo:hi], thetas[lo:hi], traj), dtype=np.float64)
            except Exception as err:    # (a refused row: the collective below must still be joined, see `_verdict`)
                local, refused = np.full(hi - lo, np.nan), err
        full = all_gather_logl_ragged(torch.from_numpy(local).to(device), sizes, self._group).cpu().numpy()
        if np.isnan(full).any():
            from . import _lib
            _lib.segments_from_st(np.asarray(ss, dtype=np.float64), np.asarray(thetas), len(traj), int(self._model.nStates))
        if refused is not None:
            raise refused
        return full


def sample_many_distributed(trajs, model, group=None, seed=None, gather='all', **kwargs):
    """
    `core.sample_many` over all ranks of a process group: whole trajectories are assigned to ranks
    (`shard_by_trajectory`, cost ~ length), every rank runs the fused inference of its own trajectories on its
    own GPU -- likelihood launches AND host-side sampler bookkeeping scale with the number of ranks, there is no
    communication while sampling -- and the results (plain picklable objects) are exchanged once at the end.

    gather : 'all' -- every rank returns the full list, in the order of ``trajs`` (``all_gather_object``; the
             results carry all samples drawn, ~100 bytes per sample: mind the size for thousands of trajectories);
             'root' -- only rank 0 does, the others get ``None`` in place of foreign results (``gather_object``);
             None -- no exchange: foreign entries are ``None``.
    seed : rank r seeds the global NumPy stream with ``seed + r`` first (the trajectories a rank gets depend on
           the number of ranks, so a run is reproducible for a fixed world size).

    Without ``torch.distributed`` initialised this is `core.sample_many`.
    """
    from .core import sample_many
    trajs = list(trajs)
    try:
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized()
    except ImportError:  # pragma: no cover
        active = False
    if not active:
        if seed is not None:
            np.random.seed(seed)
        return sample_many(trajs, model, **kwargs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    owners = shard_by_trajectory([len(t) for t in trajs], np.ones(len(trajs)), world)
    if seed is not None:
        np.random.seed(seed + rank)
    mine = owners[rank]
    local = sample_many([trajs[j] for j in mine], model, **kwargs) if len(mine) else []
    out = [None] * len(trajs)
    for j, r in zip(mine.tolist(), local):
        out[j] = r
    if gather is None:
        return out
    payload = (mine.tolist(), local)
    if gather == 'all':
        gathered = [None] * world
        dist.all_gather_object(gathered, payload, group=group)
    elif gather == 'root':
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(payload, gathered, dst=0, group=group)
        if rank != 0:
            return out
    else:
        raise ValueError("gather must be 'all', 'root' or None")
    for idx, res in gathered:
        for j, r in zip(idx, res):
            out[j] = r
    return out
