#!/usr/bin/env python3
"""
Kernel A/B experiments: build variants of libbild_amd.so with extra -D flags, then time the
same workloads with each (one process per variant, kernel time from HIP events).

    python tools/ab.py build  name1:-DFOO=1  name2:"-DBAR=2 -DBAZ"     # here (hipcc cross-compiles)
    python tools/ab.py run [--samples 10000,80000]                      # on the GPU box
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, 'bild_amd', 'variants')
CSRC = os.path.join(ROOT, 'bild_amd', 'csrc')


def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for f in os.listdir(VDIR):
        os.remove(os.path.join(VDIR, f))
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(':')
        out = os.path.join(VDIR, f'libbild_amd_{name}.so')
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-shared'] + flags.split() + \
              [os.path.join(CSRC, f) for f in ('api.cpp', 'amis_host.cpp', 'comm.cpp', 'kernels.hip', 'walk.hip', 'wide.hip', 'dense_mfma.hip',
                                               'modal_mfma.hip', 'amis_device.hip', 'schedule.hip')] + ['-o', out]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"variant {name} failed to build")
    print("built:", sorted(os.listdir(VDIR)))


def run(argv):
    samples = '10000'
    extra = []
    if '--samples' in argv:
        samples = argv[argv.index('--samples') + 1]
    if '--extra' in argv:
        extra = argv[argv.index('--extra') + 1].split()
    geoms = ['']
    if '--geoms' in argv:
        geoms = argv[argv.index('--geoms') + 1].split(',')
    variants = sorted(f for f in os.listdir(VDIR) if f.endswith('.so'))
    variants = [(v, g) for v in variants for g in geoms]
    for ns in samples.split(','):
        for v, g in variants:
            env = dict(os.environ, BILD_AMD_LIB=os.path.join(VDIR, v))
            if g != '':
                env['BILD_GEOM'] = g
            cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '30', '--warmup', '3', '--no-cpu-baseline',
                   '--no-secondary', '--samples', ns] + extra
            r = subprocess.run(cmd, env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if not line:
                print(f"{v:40s} n={ns:>7s} FAILED\n{r.stderr[-2000:]}")
                continue
            j = json.loads(line[-1])
            print(f"{v[12:-3] + ('/g' + g if g else ''):28s} n={ns:>7s} value={j['value'] / 1e6:8.3f} M/s  kernel={j['roofline']['kernel_ms'] * 1e3:9.1f} us"
                  f"  frac={j['roofline']['frac']:.3f}  seam={j['api_seam']['value'] / 1e6:7.3f} M/s", flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
