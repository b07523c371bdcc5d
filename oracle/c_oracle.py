"""ctypes binding of oracle/calib_oracle.c (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle_calib.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def bin_ids(p, thr):
    p = np.ascontiguousarray(p, dtype=np.float32).reshape(-1)
    thr = np.ascontiguousarray(thr, dtype=np.float32)
    ids = np.empty(p.size, dtype=np.uint8)
    lib().orc_bin_ids(_ptr(p), ctypes.c_size_t(p.size), _ptr(thr), ctypes.c_int(thr.size + 1), _ptr(ids))
    return ids


def ece_hist(p, target, mask, thr):
    p = np.ascontiguousarray(p, dtype=np.float32).reshape(-1)
    target = np.ascontiguousarray(target, dtype=np.uint8).reshape(-1)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
    thr = np.ascontiguousarray(thr, dtype=np.float32)
    nb = thr.size + 1
    count = np.zeros(nb, dtype=np.uint64)
    sum_conf = np.zeros(nb, dtype=np.float64)
    sum_pos = np.zeros(nb, dtype=np.uint64)
    lib().orc_ece_hist(_ptr(p), _ptr(target), _ptr(mask), ctypes.c_size_t(p.size), _ptr(thr), ctypes.c_int(nb),
                       _ptr(count), _ptr(sum_conf), _ptr(sum_pos))
    return count, sum_conf, sum_pos


def unc_counts(unc, prediction, target, mask, thresholds):
    unc = np.ascontiguousarray(unc, dtype=np.float64).reshape(-1)
    prediction = np.ascontiguousarray(prediction, dtype=np.uint8).reshape(-1)
    target = np.ascontiguousarray(target, dtype=np.uint8).reshape(-1)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    out = np.zeros((thr.size, 8), dtype=np.uint64)
    lib().orc_unc_counts(_ptr(unc), _ptr(prediction), _ptr(target), _ptr(mask), ctypes.c_size_t(unc.size),
                         _ptr(thr), ctypes.c_int(thr.size), _ptr(out))
    return out
