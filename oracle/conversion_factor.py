#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE -- BUILD CONTAINER ONLY (needs /root/reference).

SURVEY 8(d) "CPU baseline beside it": the reference's Cython kernel never leaves the build container (oracle/_ref/ is
listed in .gpurunignore), so the GPU box times this repository's own C restatement (oracle/msrouse_logl.c, "port",
bit-pinned to the reference goldens by tests/test_oracle.py) and bench.py converts that figure into a
reference-equivalent one with the factor measured HERE: both kernels on the same host core, on the same inputs -- the
bench batch (configs[1]: T = 1000, 2-state, N = 20, d = 3, k = 4), each driven by the Python loop FixedkSampler.logL runs
(bild/amis.py:735-739), BLAS pinned to one thread, interleaved in blocks so that a drifting clock hits both alike.

    python oracle/conversion_factor.py [seconds per kernel]      -> oracle/conversion_factor.json (committed)
"""
import json
import os
import platform
import sys
import time

for _v in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ[_v] = '1'

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    import helpers as H
    import bench
    from oracle import oracle
    ref = oracle.load_reference_cython()
    if ref is None:
        raise SystemExit("the reference is not present here: this script runs in the build container only")
    T, k = 1000, 4
    model, trajs, ss, thetas = bench.build_workload(0, 10000, T, k)
    traj = trajs[0]
    states = H.expand(ss[:4096], thetas[:4096], T)

    class M:
        pass
    m = M()
    m.models, m.measurement, m.d, m._get_noise = model.models, model.measurement, model.d, model._get_noise
    arrays, w, err, x = model.arrays(), model.measurement, model.localization_error, traj[:]

    def one_ref(i):
        return ref(m, H.ProfileView(states[i]), traj)

    def one_port(i):
        return oracle.logl(arrays, w, err, x, states[i])

    worst = max(abs(one_ref(i) - one_port(i)) for i in range(8))
    counts = {'reference': 0, 'port': 0}
    spent = {'reference': 0.0, 'port': 0.0}
    block = 1.0
    while min(spent.values()) < seconds:
        for name, fn in (('reference', one_ref), ('port', one_port)):
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < block:
                fn((counts[name] + n) % len(states))
                n += 1
            spent[name] += time.perf_counter() - t0
            counts[name] += n
    rates = {name: counts[name] / spent[name] for name in counts}
    cpu = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next(line.split(':', 1)[1].strip() for line in f if line.startswith('model name'))
    except Exception:
        pass
    out = {
        'what': 'reference Cython MSRouse_logL (oracle/_ref, built unmodified by oracle/build_ref.py) against the C restatement '
                '(oracle/msrouse_logl.c) on ONE core of the build container, same inputs, interleaved 1 s blocks',
        'workload': {'T': T, 'states': 2, 'N': 20, 'd': 3, 'k': k, 'profiles': len(states)},
        'reference_evals_per_s': rates['reference'], 'port_evals_per_s': rates['port'],
        'reference_over_port': rates['reference'] / rates['port'],
        'seconds_per_kernel': seconds, 'max_abs_diff_first_8': worst,
        'host': {'cpu': cpu, 'machine': platform.machine(), 'python': platform.python_version(), 'numpy': np.__version__},
        'measured': time.strftime('%Y-%m-%d'),
        'script': 'oracle/conversion_factor.py',
    }
    path = os.path.join(ROOT, 'oracle', 'conversion_factor.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
        f.write('\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
