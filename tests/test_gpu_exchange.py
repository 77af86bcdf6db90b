"""
The direct exchange (csrc/exchange_kernel.hip, bild_exchange_*) rehearsed with TWO PROCESSES SHARING THE ONE GPU: the
receive blocks are mapped into the other process through hipIpcGetMemHandle / hipIpcOpenMemHandle exactly as between two
GPUs, and the protocol -- peer stores, release, flag, bounded poll, acquire, the two parities, the step counter across its
32-bit wrap -- runs as it would over xGMI.  (What this cannot show is the cost of a hop and the caching of peer writes
between devices: no second GPU in the pool.)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, time
import numpy as np
root, base, rank, world = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
torch.cuda.set_device(0)
import helpers as H, bild_amd
from bild_amd import _lib, dist as bdist
dev = torch.device("cuda", 0)
n = 1001                                    # odd: the gathered shards are not 16-byte aligned
ex = bdist.DirectExchange.from_files(base, world, rank, n, nonce="t")
ex._x.set_step(0xFFFFFFF0)                  # the 32-bit step counter wraps inside the loop below
open(f"{base}.step.{rank}", "w").close()
while not all(os.path.exists(f"{base}.step.{r}") for r in range(world)):
    time.sleep(0.005)
recv = torch.zeros(world * n, dtype=torch.float64, device=dev)
ok = True
stream = torch.cuda.current_stream().cuda_stream
for step in range(40):
    send = torch.arange(n, dtype=torch.float64, device=dev) * (rank + 1) + 1000.0 * step
    ex.allgather(send.data_ptr(), recv.data_ptr(), n, stream)
    if step % 3 == 0:                       # some steps consumed right away, others queued back to back
        got = recv.cpu().numpy().reshape(world, n)
        ex.status()
        for r in range(world):
            ok = ok and np.array_equal(got[r], np.arange(n) * (r + 1) + 1000.0 * step)
got = recv.cpu().numpy().reshape(world, n)
ex.status()
for r in range(world):
    ok = ok and np.array_equal(got[r], np.arange(n) * (r + 1) + 1000.0 * 39)

# a peer that does not arrive: the bounded wait gives up and the status call says so (rank 1 sits this one out, and keeps its
# block mapped until rank 0 is through)
timed_out = None
if rank == 0:
    ex._x.set_timeout(0.3)
    ex.allgather(send.data_ptr(), recv.data_ptr(), n, stream)
    torch.cuda.synchronize()
    try:
        ex.status()
        timed_out = False
    except _lib.BildAmdError as err:
        timed_out = "rank 1 did not deliver" in str(err)
    open(f"{base}.timeout_done", "w").close()
else:
    while not os.path.exists(f"{base}.timeout_done"):
        time.sleep(0.005)
ok = ok and timed_out in (None, True)

# the product path: a replicated AMIS loop whose likelihood is sharded over the two processes
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 300, 2, 60), rng=rng)
ex2 = bdist.DirectExchange.from_files(base, world, rank, 64, nonce="amis")
sm = bdist.ShardedModel(model, comm=ex2)
np.random.seed(4)
sampler = bild_amd.FixedkSampler(traj, sm, k=3, N=101, max_fcomplete=0)
for _ in range(4):
    sampler.step()
np.savez(f"{base}.result.{rank}.npz", ok=ok, evidences=np.array(sampler.evidences), host_copies=sm.host_copies,
         logLs=np.concatenate([s["logLs"] for s in sampler.samples]))
print("EXCHANGE_OK" if ok else "EXCHANGE_MISMATCH")
'''


def test_two_processes_on_one_gpu(tmp_path, built_lib):
    import bild_amd
    import helpers as H
    base = str(tmp_path / 'xchg')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, '-c', _WORKER, ROOT, base, str(r), '2'], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for out, err in outs:
        assert 'EXCHANGE_OK' in out, out[-2000:] + err[-4000:]
    r0, r1 = (np.load(f"{base}.result.{r}.npz") for r in range(2))
    assert np.array_equal(r0['evidences'], r1['evidences']) and np.array_equal(r0['logLs'], r1['logLs'])   # ranks in lockstep
    assert int(r0['host_copies']) == 4                                                                      # one per step
    # ... and equal to the single-process run
    rng = np.random.default_rng(5)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 300, 2, 60), rng=rng)
    np.random.seed(4)
    ref = bild_amd.FixedkSampler(traj, model, k=3, N=101, max_fcomplete=0)
    for _ in range(4):
        ref.step()
    assert np.array_equal(np.concatenate([s['logLs'] for s in ref.samples]), r0['logLs'])
    assert np.array_equal(np.array(ref.evidences), r0['evidences'])


def test_exchange_needs_its_peers_and_world_one_is_a_copy(built_lib):
    import torch
    from bild_amd import _lib
    torch.cuda.set_device(0)
    a = _lib.ExchangeHandle(2, 0, 16)
    send = torch.ones(16, dtype=torch.float64, device='cuda')
    recv = torch.zeros(32, dtype=torch.float64, device='cuda')
    with pytest.raises(_lib.BildAmdError, match="connect"):
        a.allgather(send.data_ptr(), recv.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(_lib.BildAmdError, match="slots"):
        one = _lib.ExchangeHandle(1, 0, 8)
        one.allgather(send.data_ptr(), recv.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
    one = _lib.ExchangeHandle(1, 0, 16)        # world = 1 needs no connect: the exchange is a copy through the own block
    for step in range(3):
        send.fill_(float(step))
        one.allgather(send.data_ptr(), recv.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        one.status()
        assert np.array_equal(recv[:16].cpu().numpy(), np.full(16, float(step)))
