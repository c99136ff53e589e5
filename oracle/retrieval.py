"""ctypes front of oracle/topk_ref.c (exhaustive inner-product search restated; see the header of that file) plus the
NumPy restatement of the generate job's file writer.  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_cbuild", "libsimx_oracle.so")
_lib = None


def build():
    """gcc the C restatement (called from __graft_entry__.build(); the .so travels to the GPU box)."""
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    src = os.path.join(_HERE, "topk_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", _SO, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.simx_oracle_scores.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.simx_oracle_topk.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def scores(q, c):
    """[nq,H] x [nc,H] -> [nq,nc] float32, one fmaf chain per score in ascending h."""
    q, c = np.ascontiguousarray(q, np.float32), np.ascontiguousarray(c, np.float32)
    out = np.empty((q.shape[0], c.shape[0]), np.float32)
    _load().simx_oracle_scores(q.shape[0], c.shape[0], q.shape[1], q.ctypes.data, c.ctypes.data, out.ctypes.data)
    return out


def topk(s, k, ids=None, id_base=0):
    """k best (score desc, id asc) of each row -> (scores [nq,k] f32, ids [nq,k] i64); short rows padded (-inf, -1)."""
    s = np.ascontiguousarray(s, np.float32)
    nq, m = s.shape
    out_s, out_i = np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, np.int64)
        idp = ids.ctypes.data
    _load().simx_oracle_topk(nq, m, s.ctypes.data, idp, int(id_base), int(k), out_s.ctypes.data, out_i.ctypes.data)
    return out_s, out_i


def search(q, c, k, id_base=0):
    """IndexFlatIP.search restated: exhaustive scores then top-k."""
    return topk(scores(q, c), k, None, id_base)
