"""
Fusing the AMIS batches of many concurrently running inference loops into single launches.

The reference runs ``bild.sample`` one trajectory at a time, and inside it one likelihood
evaluation at a time.  On a GPU a single AMIS step of the default size (N = 100 profiles) is
far too small to fill the device, but the steps of *different* trajectories (and of different
k) are independent.  `run_batched` therefore runs one unmodified `core.sample` loop per
trajectory as a cooperative task and, whenever every task is waiting for likelihoods, evaluates
all pending batches with ONE call of ``model.logL_segments`` over the device-resident
trajectory set (one launch; samples carry a trajectory id).

Tasks are Python threads that are only ever run one at a time, in a fixed order, handing a
baton back and forth with the coordinator -- i.e. coroutines.  Execution is deterministic: the
global NumPy random stream is consumed in the same order on every run.
"""
import threading

import numpy as np

from .profiles import segments_from_st

_INT32_MAX = np.iinfo(np.int32).max


class _Cancelled(Exception):
    """raised inside a task whose run was abandoned because another task failed"""


class _Task:
    def __init__(self, index, traj):
        self.index = index
        self.traj = traj
        self.go = threading.Event()       # coordinator -> task: run until you need likelihoods
        self.parked = threading.Event()   # task -> coordinator: I am waiting (or finished)
        self.request = None               # (seg_start, seg_state) while waiting
        self.answer = None
        self.result = None
        self.error = None
        self.done = False


class BatchingModel:
    """
    What a task sees instead of the real model: same attributes, but ``logL_st_batch`` parks the
    task until the coordinator has evaluated the fused batch.
    """

    def __init__(self, model, task, coordinator):
        self._model = model
        self._task = task
        self._coordinator = coordinator
        self.transitions = model.transitions

    @property
    def nStates(self):
        return self._model.nStates

    @property
    def d(self):
        return self._model.d

    def logL(self, profile, traj):
        return self._model.logL(profile, traj)

    def logL_st_batch(self, ss, thetas, traj):
        seg_start, seg_state = segments_from_st(ss, thetas, len(traj))
        task = self._task
        task.request = (seg_start, seg_state)
        task.parked.set()
        task.go.wait()
        task.go.clear()
        out, task.answer = task.answer, None
        if isinstance(out, _Cancelled):
            raise out
        return out


def run_batched(trajs, model, loop, return_exceptions=False, **kwargs):
    """
    Run ``loop(traj, model, **kwargs)`` (normally `core.sample`) for every trajectory, fusing
    their likelihood batches.  ``model`` must offer ``logL_segments(seg_start, seg_state, trajs,
    traj_id)`` (`models.MultiStateRouse` does).

    An exception inside one loop (the samplers raise e.g. ``RuntimeError`` when a proposal fit does
    not converge, amis.py:441) is re-raised here after the other loops have been unwound; with
    ``return_exceptions=True`` it becomes that trajectory's entry of the returned list instead and
    the other loops run to completion.
    """
    trajs = list(trajs)
    tasks = [_Task(i, t) for i, t in enumerate(trajs)]

    def body(task):
        task.go.wait()
        task.go.clear()
        try:
            task.result = loop(task.traj, BatchingModel(model, task, None), **kwargs)
        except BaseException as err:  # propagate to the caller of run_batched
            task.error = err
        task.done = True
        task.parked.set()

    threads = [threading.Thread(target=body, args=(t,), daemon=True) for t in tasks]
    for th in threads:
        th.start()

    live = list(tasks)
    while live:
        # run every live task, one at a time and in order, up to its next likelihood request
        for task in live:
            task.parked.clear()
            task.go.set()
            task.parked.wait()
        failed = [t for t in live if t.error is not None]
        if failed and not return_exceptions:
            for task in live:   # unwind the loops that are parked on a request
                if not task.done:
                    task.answer = _Cancelled()
                    task.parked.clear()
                    task.go.set()
                    task.parked.wait()
            for th in threads:
                th.join()
            raise failed[0].error
        live = [t for t in live if not t.done]
        if not live:
            break
        # one fused evaluation for everything that is pending
        K1 = max(t.request[0].shape[1] for t in live)
        starts, states, tid = [], [], []
        for t in live:
            a, b = t.request
            n, k1 = a.shape
            if k1 < K1:   # pad with empty segments (start beyond any trajectory)
                a = np.concatenate([a, np.full((n, K1 - k1), _INT32_MAX, dtype=np.int32)], axis=1)
                b = np.concatenate([b, np.repeat(b[:, -1:], K1 - k1, axis=1)], axis=1)
            starts.append(a)
            states.append(b)
            tid.append(np.full(n, t.index, dtype=np.int32))
        out = model.logL_segments(np.concatenate(starts), np.concatenate(states), trajs, np.concatenate(tid))
        pos = 0
        for t in live:
            n = t.request[0].shape[0]
            t.answer = np.asarray(out[pos:pos + n], dtype=np.float64)
            t.request = None
            pos += n
    for th in threads:
        th.join()
    return [t.result if t.error is None else t.error for t in tasks]
