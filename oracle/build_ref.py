#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE -- builds the *real* reference kernel, unmodified, for use as checker
and CPU baseline.  Nothing under bild_amd/ may import or execute anything under oracle/.

Compiles the reference's one native component, bild/src/MSRouse_logL.pyx, from where it
lies under /root/reference into ``oracle/_ref/MSRouse_logL.<abi>.so``:

    cython  (pyx -> C, into a temporary directory that is deleted afterwards)
    gcc -O2 (the reference's own setup.py:42-56 uses default setuptools flags, i.e. -O2,
             with the macro NPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION)

No reference source is copied into the repository: the only output is the shared object,
and ``oracle/_ref/`` is git-ignored.  The reference's own build system (setup.py /
Makefile) is not run.  The binary stays in the build container: ``oracle/_ref/`` is listed in
``.gpurunignore`` too (the reference never leaves this container).  Where
/root/reference is absent (the GPU box) this script is a no-op and returns None; the CPU
baseline there is the C restatement, converted with oracle/conversion_factor.json.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PYX = '/root/reference/bild/src/MSRouse_logL.pyx'
OUT_DIR = os.path.join(HERE, '_ref')


def so_path():
    return os.path.join(OUT_DIR, 'MSRouse_logL' + sysconfig.get_config_var('EXT_SUFFIX'))


def build(force=False, verbose=True):
    out = so_path()
    if not os.path.exists(REF_PYX):
        if verbose:
            print(f"[oracle/_ref] reference not present; using prebuilt {out}"
                  f" ({'found' if os.path.exists(out) else 'MISSING'})")
        return out if os.path.exists(out) else None
    if os.path.exists(out) and not force and os.path.getmtime(out) >= os.path.getmtime(REF_PYX):
        return out

    import numpy as np
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='bild_ref_build_')
    try:
        c_file = os.path.join(tmp, 'MSRouse_logL.c')
        subprocess.check_call([sys.executable, '-m', 'cython', '-3', REF_PYX, '-o', c_file],
                              env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
        cmd = ['gcc', '-shared', '-fPIC', '-O2', '-w',
               '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION',
               '-I' + sysconfig.get_paths()['include'],
               '-I' + np.get_include(),
               c_file, '-o', out]
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if verbose:
        print(f"[oracle/_ref] built {out}")
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
