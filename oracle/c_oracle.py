"""TEST INFRASTRUCTURE ONLY -- ctypes loader of oracle/liboracle.so (the plain-C restatement)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_P = ctypes.c_void_p


def load():
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.isfile(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    lib.oracle_num_threads.restype = ctypes.c_int
    lib.oracle_sim.argtypes = [_P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_float, _P]
    lib.oracle_softmax_ce.restype = ctypes.c_double
    lib.oracle_softmax_ce.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int64, ctypes.c_float, _P, _P, _P]
    lib.oracle_dq.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]
    lib.oracle_dc.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]
    lib.oracle_rank_of_gold.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int64, _P]
    lib.oracle_train_step.restype = ctypes.c_double
    lib.oracle_train_step.argtypes = [_P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int64, _P,
                                      ctypes.c_float, ctypes.c_float, _P, _P, _P, _P]
    return lib


def ptr(a):
    return a.ctypes.data_as(_P) if a is not None else None


def train_step(lib, Q, C, y, y_offset, mask, T, Nq_global):
    """One rank's step with the C port.  Returns dict(loss_sum, S, G, dQ, dC)."""
    Q = np.ascontiguousarray(Q, np.float32)
    C = np.ascontiguousarray(C, np.float32)
    y = np.ascontiguousarray(y, np.int64)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    B, d = Q.shape
    Nc = C.shape[0]
    S = np.empty((B, Nc), np.float32)
    G = np.empty((B, Nc), np.float32)
    dQ = np.empty((B, d), np.float32)
    dC = np.empty((Nc, d), np.float32)
    tot = lib.oracle_train_step(ptr(Q), B, ptr(C), Nc, d, ptr(y), int(y_offset), ptr(m), 1.0 / T, 1.0 / (T * Nq_global),
                                ptr(S), ptr(G), ptr(dQ), ptr(dC))
    return dict(loss_sum=tot, S=S, G=G, dQ=dQ, dC=dC)
