#!/usr/bin/env python3
"""
Golden vectors for `ChoiceSampler`, from the REFERENCE's
bild/choicesampler.py (imported through oracle/ref_loader.py; build container only) with a
fixed ``np.random.seed``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_choice_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_loader  # noqa: E402

rcs = ref_loader.load('choicesampler')
assert rcs is not None

CASES = {
    'peaked': dict(muhat=[-30., -12., -10.5, -10.2, -11., -13.], se=[1e-5, 0.3, 0.2, 0.25, 0.4, 0.5],
                   N=[np.inf, 20, 25, 20, 20, 20], dE=0., seed=21, omit=[4, 5]),
    'margin': dict(muhat=[-20., -11., -10., -9.6, -9.5, -9.45, -9.7], se=[1e-5, 0.1, 0.2, 0.2, 0.3, 0.3, 0.3],
                   N=[np.inf, 30, 20, 20, 20, 20, 20], dE=1.0, seed=22, omit=[5, 6]),
    'flat': dict(muhat=[-5., -5.1, -4.9, -5.05], se=[0.5, 0.5, 0.5, 0.5], N=[5, 5, 5, 5], dE=0.5, seed=23, omit=[3]),
}

if __name__ == '__main__':
    out = {}
    for name, c in CASES.items():
        np.random.seed(c['seed'])
        cs = rcs.ChoiceSampler(np.array(c['muhat']), np.array(c['se']) ** 2, np.array(c['N'], dtype=float), c['dE'],
                               samplesize=4000)
        out[f'{name}_muhat'] = np.array(c['muhat'])
        out[f'{name}_shat'] = np.array(c['se']) ** 2
        out[f'{name}_N'] = np.array(c['N'], dtype=float)
        out[f'{name}_dE'] = c['dE']
        out[f'{name}_seed'] = c['seed']
        out[f'{name}_omit'] = np.array(c['omit'])
        out[f'{name}_n0'] = cs.n0
        out[f'{name}_bestk'] = cs.bestk
        out[f'{name}_Dn'] = cs.Dn()
        out[f'{name}_KLD'] = cs.KLD_moreSamples()
        out[f'{name}_Ila'] = cs.KLD_omitK(np.array(c['omit']))
        print(name, cs.n0, out[f'{name}_KLD'], out[f'{name}_Ila'])
    np.savez_compressed(os.path.join(HERE, 'choicesampler.npz'), **out)
