"""Builds librcu_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'librcu_hip.so')


def build(force=False, jobs=4):
    if force:
        subprocess.check_call(['make', '-s', '-C', CSRC, 'clean'])
    subprocess.check_call(['make', '-s', '-j{}'.format(jobs), '-C', CSRC])
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('hipcc build did not produce {}'.format(LIB_PATH))
    return LIB_PATH


if __name__ == '__main__':
    print(build())
