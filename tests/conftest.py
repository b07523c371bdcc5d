import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_state(g):
    return {k[len('sd::'):]: v for k, v in g.items() if k.startswith('sd::')}


def golden_params(g):
    return eval(str(g['params']), {'__builtins__': {}}, {'dict': dict})  # repr(dict) written by the generator


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
