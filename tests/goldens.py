""" loader for the committed golden vectors (tests/golden/*.npz) """
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def names():
    """ the kernel-level goldens (the AMIS / ChoiceSampler fixtures live beside them under their own prefixes) """
    all_names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, 'golden', '*.npz')))
    return [n for n in all_names if not n.startswith(('amis_', 'choicesampler', 'st2profile'))]


def load(name):
    z = np.load(os.path.join(HERE, 'golden', name + '.npz'))
    g = {k: z[k] for k in z.files}
    g['states'] = g['states'].astype(np.int64)
    g['arrays'] = {k: g[k] for k in ('B', 'G', 'Sig', 'M0', 'C0')}
    return g
