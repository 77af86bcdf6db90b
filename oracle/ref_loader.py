"""
TEST INFRASTRUCTURE (build container only) -- import the reference's pure-Python modules
(amis, util, choicesampler, postproc) from /root/reference without executing its package
``__init__`` (which needs the absent `rouse` / `noctiluca`), as SURVEY.md section 8c describes:
a synthetic package object named ``bild`` whose ``__path__`` points at the reference tree.

Used only by tests/golden/make_*_golden.py to GENERATE fixtures; never on the GPU box,
never by the product.
"""
import importlib
import os
import sys
import types

REF = '/root/reference/bild'


def available():
    return os.path.isdir(REF)


def load(*names):
    """ e.g. load('amis') -> reference bild.amis module object """
    if not available():
        return None
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        if 'bild' not in sys.modules or not hasattr(sys.modules['bild'], '_bild_amd_synthetic'):
            pkg = types.ModuleType('bild')
            pkg.__path__ = [REF]
            pkg._bild_amd_synthetic = True
            sys.modules['bild'] = pkg
        mods = [importlib.import_module('bild.' + n) for n in names]
    finally:
        sys.dont_write_bytecode = old
    return mods[0] if len(mods) == 1 else mods
