"""ctypes front door to the CPU oracle (oracle/vdl_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never from ``mplan2vdl_amd`` (the product).
Parity status: see the header of vdl_oracle.c ("parity unpinned" against the original
Voodoo backend; pinned by README golden lines + independent SQL evaluators).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvdl_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "vdl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvdl_oracle.so"])
    return _SO


class ColSpec(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("col_id", ctypes.c_uint64),
                ("lo", ctypes.c_int64), ("hi", ctypes.c_int64), ("mul", ctypes.c_int64), ("add", ctypes.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("VDL_ORACLE_SO")          # e.g. the ASan build of tools/sanitize/run.sh
        if not so:
            so = build()
        L = ctypes.CDLL(so)
        L.orc_open.restype = ctypes.c_void_p
        L.orc_close.argtypes = [ctypes.c_void_p]
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_last_error.argtypes = [ctypes.c_void_p]
        L.orc_last_run_seconds.restype = ctypes.c_double
        L.orc_last_run_seconds.argtypes = [ctypes.c_void_p]
        L.orc_last_ops.restype = ctypes.c_int64
        L.orc_last_ops.argtypes = [ctypes.c_void_p]
        L.orc_add_column.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]
        L.orc_run.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.orc_n_outputs.argtypes = [ctypes.c_void_p]
        L.orc_output.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                 ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.POINTER(ctypes.c_int64)),
                                 ctypes.POINTER(ctypes.c_int64)]
        L.orc_col_id.restype = ctypes.c_uint64
        L.orc_col_id.argtypes = [ctypes.c_char_p]
        L.orc_gen_column.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                     ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.orc_sql_q6.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int,
                                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.orc_sql_q6_generated.argtypes = [ctypes.POINTER(ColSpec), ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.orc_sql_q1.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                                         ctypes.POINTER(ctypes.c_int)]
        L.orc_sql_q1_generated.argtypes = [ctypes.POINTER(ColSpec), ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_keep_vectors.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_vector.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                 ctypes.POINTER(ctypes.POINTER(ctypes.c_int64)), ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                 ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class Oracle:
    """Scalar CPU VDL interpreter: ``add_column`` host arrays, then ``run(vdl_text)``.

    ``run`` returns the same JSON-shaped dict the reference pipeline expects from its
    executor (/root/reference/resolve.py:8-32): ``{"results": {tmpN: {".name": [...]}},
    "timings": {...}}``.
    """

    def __init__(self):
        self._L = lib()
        self._c = ctypes.c_void_p(self._L.orc_open())
        self._keep = {}

    def close(self):
        if self._c:
            self._L.orc_close(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_column(self, name, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.kind != "i" or arr.dtype.itemsize not in (1, 2, 4, 8):
            raise OracleError("columns must be signed integer arrays")
        self._keep[name] = arr
        rc = self._L.orc_add_column(self._c, name.encode(), arr.ctypes.data_as(ctypes.c_void_p),
                                    arr.dtype.itemsize, arr.shape[0])
        if rc:
            raise OracleError(self._L.orc_last_error(self._c).decode())

    def run(self, vdl_text):
        data = vdl_text.encode() if isinstance(vdl_text, str) else vdl_text
        rc = self._L.orc_run(self._c, data, len(data))
        if rc:
            msg = self._L.orc_last_error(self._c).decode()
            if isinstance(vdl_text, str):               # evidence for DESIGN.md section 8 (9): did the bytes handed over change?
                fresh = vdl_text.encode()
                changed = [(i, data[i], fresh[i]) for i in range(min(len(data), len(fresh))) if data[i] != fresh[i]][:16]
                if changed:
                    msg += "  [the bytes handed to orc_run differ from their source str at (offset, is, was) %r]" % (changed,)
            raise OracleError(msg)
        results = {}
        for k in range(self._L.orc_n_outputs(self._c)):
            name, tmp = ctypes.c_char_p(), ctypes.c_char_p()
            vals, n = ctypes.POINTER(ctypes.c_int64)(), ctypes.c_int64()
            self._L.orc_output(self._c, k, ctypes.byref(name), ctypes.byref(tmp), ctypes.byref(vals), ctypes.byref(n))
            arr = np.ctypeslib.as_array(vals, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64)
            results[tmp.value.decode()] = {"." + name.value.decode(): arr.tolist()}
        secs = self._L.orc_last_run_seconds(self._c)
        return {"results": results,
                "timings": {"timeInMicrosecondsForCpuOracle": int(secs * 1e6)}}

    def keep_vectors(self, keep=True):
        """Keep every statement's vector readable after run() (statement-by-statement comparison with the engine)."""
        self._L.orc_keep_vectors(self._c, int(bool(keep)))

    def vector(self, node_id):
        """(values, holds_value) of statement `node_id` after a run with keep_vectors: two arrays over the n slots
        (int64 / bool), or None if the statement made no vector."""
        n, frm, step = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        vals, ok = ctypes.POINTER(ctypes.c_int64)(), ctypes.POINTER(ctypes.c_uint8)()
        if self._L.orc_vector(self._c, int(node_id), ctypes.byref(n), ctypes.byref(vals), ctypes.byref(ok), ctypes.byref(frm), ctypes.byref(step)):
            return None
        k = n.value
        if vals:
            v = np.ctypeslib.as_array(vals, shape=(k,)).copy() if k else np.zeros(0, np.int64)
        else:
            v = (np.uint64(frm.value & (2**64 - 1)) + np.arange(k, dtype=np.uint64) * np.uint64(step.value & (2**64 - 1))).astype(np.int64)
        o = np.ctypeslib.as_array(ok, shape=(k,)).astype(bool) if (ok and k) else np.ones(k, bool)
        return v, o

    @property
    def last_seconds(self):
        return self._L.orc_last_run_seconds(self._c)


def col_id(name):
    return int(lib().orc_col_id(name.encode()))


def gen_column(dtype, row0, n, seed, cid, lo, hi, mul=1, add=0):
    out = np.empty(n, dtype=dtype)
    rc = lib().orc_gen_column(out.ctypes.data_as(ctypes.c_void_p), out.dtype.itemsize, row0, n, seed, cid, lo, hi, mul, add)
    if rc:
        raise OracleError("orc_gen_column failed")
    return out


def sql_q6(shipdate, discount, quantity, extprice, threads=1):
    """Fused scalar loop written from the Q6 SQL text; returns (revenue, n_selected)."""
    a = [np.ascontiguousarray(shipdate, np.int32), np.ascontiguousarray(discount, np.int64),
         np.ascontiguousarray(quantity, np.int64), np.ascontiguousarray(extprice, np.int64)]
    rev, cnt = ctypes.c_int64(), ctypes.c_int64()
    lib().orc_sql_q6(*[x.ctypes.data_as(ctypes.c_void_p) for x in a], a[0].shape[0], threads,
                     ctypes.byref(rev), ctypes.byref(cnt))
    return rev.value, cnt.value


def sql_q6_generated(specs, row0, n, threads=1):
    """Q6 over generated rows without storing columns; specs = 4 (seed, col_id, lo, hi, mul, add)
    tuples in the order shipdate, discount, quantity, extendedprice."""
    arr = (ColSpec * 4)(*[ColSpec(*s) for s in specs])
    rev, cnt = ctypes.c_int64(), ctypes.c_int64()
    lib().orc_sql_q6_generated(arr, row0, n, threads, ctypes.byref(rev), ctypes.byref(cnt))
    return rev.value, cnt.value


def sql_q1(shipdate, returnflag, linestatus, quantity, extprice, discount, tax):
    """Fused scalar loop written from the Q1 SQL text.  Returns an int64 array [groups, 10]:
    rf, ls, sum_qty, sum_base_price, sum_disc_price, sum_charge, avg_qty, avg_price, avg_disc, count."""
    a = [np.ascontiguousarray(shipdate, np.int32), np.ascontiguousarray(returnflag, np.int32),
         np.ascontiguousarray(linestatus, np.int32), np.ascontiguousarray(quantity, np.int64),
         np.ascontiguousarray(extprice, np.int64), np.ascontiguousarray(discount, np.int64),
         np.ascontiguousarray(tax, np.int64)]
    out = np.zeros((64, 10), np.int64)
    ng = ctypes.c_int()
    rc = lib().orc_sql_q1(*[x.ctypes.data_as(ctypes.c_void_p) for x in a], a[0].shape[0],
                          out.ctypes.data_as(ctypes.c_void_p), 64, ctypes.byref(ng))
    if rc:
        raise OracleError("orc_sql_q1 failed")
    return out[:ng.value].copy()


def sql_q1_generated(specs, row0, n, threads=1):
    """Q1 over generated rows; specs = 7 (seed, col_id, lo, hi, mul, add) tuples in the order shipdate,
    returnflag, linestatus, quantity, extendedprice, discount, tax.  Same layout as sql_q1."""
    arr = (ColSpec * 7)(*[ColSpec(*s) for s in specs])
    out = np.zeros((64, 10), np.int64)
    ng = ctypes.c_int()
    rc = lib().orc_sql_q1_generated(arr, row0, n, threads, out.ctypes.data_as(ctypes.c_void_p), 64, ctypes.byref(ng))
    if rc:
        raise OracleError("orc_sql_q1_generated failed")
    return out[:ng.value].copy()


def max_threads():
    return int(lib().orc_max_threads())
