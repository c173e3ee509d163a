"""Host-side mirror of the reference's executor interface.

In the reference the executor is reached as text over a pipe: VDL in, JSON out
(/root/reference/eval_query.sh:18-26; reply shape /root/reference/resolve.py:8-32).
``Engine.run_vdl(text)`` is that request/response pair as a function call; everything it
does goes through the C ABI of libvdl.so (include/vdl.h) into hand-written HIP kernels.
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from . import datagen


class VdlError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("[vdl error %d] %s" % (code, message))
        self.code = code


class DeviceValues:
    """An output left in device memory (Plan.set_device_outputs): int64 values at `ptr`, valid until the plan runs again."""

    def __init__(self, ptr, n, owner):
        self.ptr, self.n, self._owner = int(ptr), int(n), owner

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.n,), "typestr": "<i8", "data": (self.ptr, False), "version": 3, "strides": None}

    def __len__(self):
        return self.n


class Plan:
    def __init__(self, engine, handle, text):
        self._e = engine
        self._h = handle
        self.text = text

    def close(self):
        if self._h:
            self._e._L.vdl_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self):
        return self._e._L.vdl_plan_describe(self._h).decode()

    @property
    def is_fused(self):
        return bool(self._e._L.vdl_plan_is_fused(self._h))

    def set_fusion(self, enabled):
        self._e._check(self._e._L.vdl_plan_set_fusion(self._h, int(bool(enabled))))

    def set_profiling(self, enabled):
        self._e._check(self._e._L.vdl_plan_set_profiling(self._h, int(bool(enabled))))

    def set_jit(self, enabled, tune=False):
        """Scan kernels specialised for this plan by hiprtc at the next run; tune: rows per lane chosen by timing at that
        run (vdl.h: vdl_plan_set_jit)."""
        self._e._check(self._e._L.vdl_plan_set_jit(self._h, (2 if tune else 1) if enabled else 0))

    def jit_note(self):
        return self._e._L.vdl_plan_jit_note(self._h).decode()

    def jit_check(self):
        """Build (not load, not run: no GPU needed) the specialised kernels against the registered columns; the note."""
        self._e._check(self._e._L.vdl_plan_jit_check(self._e._c, self._h))
        return self.jit_note()

    def set_trace(self, enabled):
        """Keep a host copy of every statement's vector (statement-by-statement runs only): see `traced()`."""
        self._e._check(self._e._L.vdl_plan_set_trace(self._h, int(bool(enabled))))

    def traced(self):
        """[(statement id, form, n, values or None, holds_value or None)] of the last traced run, in execution order."""
        L, out = self._e._L, []
        for k in range(L.vdl_n_traced(self._h)):
            node, form, n = ctypes.c_int(), ctypes.c_char_p(), ctypes.c_int64()
            vals, ok = ctypes.POINTER(ctypes.c_int64)(), ctypes.POINTER(ctypes.c_uint8)()
            L.vdl_traced(self._h, k, ctypes.byref(node), ctypes.byref(form), ctypes.byref(n), ctypes.byref(vals), ctypes.byref(ok))
            if ok or n.value == 0:      # null pointers = not evaluated at that point (an empty vector has nothing to point at either)
                v = np.ctypeslib.as_array(vals, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64)
                o = np.ctypeslib.as_array(ok, shape=(n.value,)).astype(bool) if n.value else np.zeros(0, bool)
            else:
                v = o = None
            out.append((node.value, form.value.decode(), n.value, v, o))
        return out

    def set_device_outputs(self, enabled):
        """Large outputs (>= 65536 values) stay in HBM: `collect()` returns them as `DeviceValues` (device pointer +
        length, `__cuda_array_interface__`: `torch.as_tensor(v, device=...)` wraps them without a copy).  They belong
        to the plan until its next run."""
        self._e._check(self._e._L.vdl_plan_set_device_outputs(self._h, int(bool(enabled))))

    def _collect(self, as_numpy=False):
        L = self._e._L
        results = {}
        for k in range(L.vdl_n_outputs(self._h)):
            name, tmp = ctypes.c_char_p(), ctypes.c_char_p()
            vals, n = ctypes.POINTER(ctypes.c_int64)(), ctypes.c_size_t()
            L.vdl_output(self._h, k, ctypes.byref(name), ctypes.byref(tmp), ctypes.byref(vals), ctypes.byref(n))
            if n.value and not vals:
                dev = ctypes.POINTER(ctypes.c_int64)()
                L.vdl_output_device(self._h, k, ctypes.byref(dev), ctypes.byref(n))
                arr = DeviceValues(ctypes.cast(dev, ctypes.c_void_p).value, n.value, self)
            elif as_numpy:
                arr = np.ctypeslib.as_array(vals, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64)
            else:
                arr = np.ctypeslib.as_array(vals, shape=(n.value,)).tolist() if n.value else []
            results[tmp.value.decode()] = {"." + name.value.decode(): arr}
        timings = {}
        for k in range(L.vdl_n_timings(self._h)):
            label, us = ctypes.c_char_p(), ctypes.c_double()
            L.vdl_timing(self._h, k, ctypes.byref(label), ctypes.byref(us))
            timings[label.value.decode()] = int(round(us.value))
        return {"results": results, "timings": timings}

    def execute(self):
        """vdl_run only: the outputs stay in the plan (borrowed until the next run); `collect()` converts them."""
        self._e._check(self._e._L.vdl_run(self._e._c, self._h))

    def collect(self, as_numpy=False):
        return self._collect(as_numpy)

    def run(self, as_numpy=False):
        """Execute on the GPU; returns {"results": {tmpN: {".name": [ints]}}, "timings": {...}} (the reply shape of
        /root/reference/resolve.py:8-32).  as_numpy=True keeps the value lists as int64 arrays: converting millions
        of result rows to Python ints costs far more than computing them."""
        self._e._check(self._e._L.vdl_run(self._e._c, self._h))
        return self._collect(as_numpy)

    # ---- sharded execution (one process per GPU) ----
    def partial_spec(self):
        n, ops = ctypes.c_int64(), ctypes.POINTER(ctypes.c_int32)()
        self._e._check(self._e._L.vdl_plan_partial_spec(self._h, ctypes.byref(n), ctypes.byref(ops)))
        return n.value, [int(ops[i]) for i in range(n.value)]

    def run_local(self, dev_ptr):
        self._e._check(self._e._L.vdl_run_local(self._e._c, self._h, ctypes.c_void_p(dev_ptr)))

    def finalize(self, dev_ptr):
        self._e._check(self._e._L.vdl_finalize(self._e._c, self._h, ctypes.c_void_p(dev_ptr)))
        return self._collect()

    def set_sharded_table(self, table):
        """Placement for plans that do not fuse: `table` is split by rows over the ranks, the others are replicated
        (then partial_spec / run_local / finalize also serve general plans whose outputs hang off global folds)."""
        self._e._check(self._e._L.vdl_plan_set_sharded_table(self._h, table.encode() if table else None))

    def set_row_offset(self, row0):
        self._e._check(self._e._L.vdl_plan_set_row_offset(self._h, int(row0)))

    def resolve_first(self, dev_ptr):
        self._e._check(self._e._L.vdl_resolve_first(self._e._c, self._h, ctypes.c_void_p(dev_ptr)))

    # ---- sharded Partition exchange (joins / sparse GROUP BY) ----
    def exchange_columns(self, sharded_table=None):
        n = ctypes.c_int()
        tbl = sharded_table.encode() if sharded_table else None
        self._e._check(self._e._L.vdl_exchange_spec(self._h, tbl, ctypes.byref(n)))
        return n.value

    def exchange_begin(self, world):
        counts = (ctypes.c_int64 * world)()
        self._e._check(self._e._L.vdl_exchange_begin(self._e._c, self._h, world, counts))
        return list(counts)

    def exchange_pack(self, dev_ptr):
        self._e._check(self._e._L.vdl_exchange_pack(self._e._c, self._h, ctypes.c_void_p(dev_ptr)))

    def exchange_finish(self, dev_ptr, n_recv, as_numpy=False):
        self._e._check(self._e._L.vdl_exchange_finish(self._e._c, self._h, ctypes.c_void_p(dev_ptr or 0), int(n_recv)))
        return self._collect(as_numpy)

    # ---- multi-GPU behind the C ABI (Engine.comm_init_*): the collectives happen inside libvdl ----
    def run_sharded(self, as_numpy=False):
        """Local phase over this rank's rows, collectives, finalisation (vdl_run_sharded).  Fold plans: every rank gets the
        full result; plans with a Partition: this rank's slice (the ranks' slices concatenate in rank order)."""
        self._e._check(self._e._L.vdl_run_sharded(self._e._c, self._h))
        return self._collect(as_numpy)

    def sharded_route(self):
        """("fold" | "set" | "exchange", every rank ends with the whole answer?) -- vdl_plan_sharded_route; raises when the plan
        has no sharded route under the placement named with set_sharded_table."""
        name, whole = ctypes.c_char_p(), ctypes.c_int()
        self._e._check(self._e._L.vdl_plan_sharded_route(self._e._c, self._h, ctypes.byref(name), ctypes.byref(whole)))
        return name.value.decode(), bool(whole.value)

    def execute_sharded(self):
        """vdl_run_sharded only: the outputs stay in the plan (`collect()` converts them)."""
        self._e._check(self._e._L.vdl_run_sharded(self._e._c, self._h))

    def run_sharded_begin(self, slot):
        self._e._check(self._e._L.vdl_run_sharded_begin(self._e._c, self._h, int(slot)))

    def run_sharded_end(self, slot):
        self._e._check(self._e._L.vdl_run_sharded_end(self._e._c, self._h, int(slot)))
        return self._collect()

    def finalize_begin(self, dev_ptr, slot):
        self._e._check(self._e._L.vdl_finalize_begin(self._e._c, self._h, ctypes.c_void_p(dev_ptr), int(slot)))

    def finalize_end(self, slot):
        self._e._check(self._e._L.vdl_finalize_end(self._e._c, self._h, int(slot)))
        return self._collect()

    def scan_stats(self):
        rows, nbytes, us = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        self._e._check(self._e._L.vdl_plan_scan_stats(self._h, ctypes.byref(rows), ctypes.byref(nbytes), ctypes.byref(us)))
        return rows.value, nbytes.value, us.value

    def scan_traffic(self):
        """(HBM bytes one launch of the dominant fused scan moves, per-column detail): vdl_plan_scan_traffic -- measurement,
        call after a run and outside timed regions (a staged scan's late columns are counted by a census launch)."""
        b, detail = ctypes.c_int64(), ctypes.c_char_p()
        self._e._check(self._e._L.vdl_plan_scan_traffic(self._e._c, self._h, ctypes.byref(b), ctypes.byref(detail)))
        return b.value, (detail.value or b"").decode()


class Engine:
    """One context = one GPU (``device=None``: host-only, can parse/describe but not run)."""

    def __init__(self, device=0):
        self._L = _lib.load()
        c = ctypes.c_void_p()
        rc = self._L.vdl_open(ctypes.byref(c), -1 if device is None else int(device))
        self._c = c
        self._keep = {}
        self._plans = weakref.WeakSet()
        if rc:
            msg = self._L.vdl_last_error(c).decode()
            self._L.vdl_close(c)
            self._c = None
            raise VdlError(rc, msg)

    def close(self):
        if self._c:
            for p in list(self._plans):     # plans hold device events / pinned buffers of this context
                p.close()
            self._L.vdl_close(self._c)
            self._c = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise VdlError(rc, self._L.vdl_last_error(self._c).decode())

    def version(self):
        return self._L.vdl_version().decode()

    def set_stream(self, hip_stream_handle):
        """Launch on this HIP stream handle (0 = the legacy default stream torch uses by default)."""
        self._check(self._L.vdl_set_stream(self._c, ctypes.c_void_p(hip_stream_handle or 0)))

    def use_torch_stream(self):
        """Launch on torch's current stream, so torch ops / collectives order against the engine."""
        import torch

        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def use_own_stream(self):
        self._check(self._L.vdl_use_own_stream(self._c))

    # ---- communicator (one process per GPU; see include/vdl.h "multi-GPU behind this boundary") ----
    def comm_unique_id(self):
        """128 bytes made by ONE rank (ncclGetUniqueId); hand them to the other ranks over any channel."""
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        self._check(self._L.vdl_comm_unique_id(buf))
        return buf.raw

    def comm_init_rccl(self, rank, world, unique_id):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise VdlError(_lib.VDL_ERR_ARG, "the communicator id is %d bytes" % _lib.COMM_ID_BYTES)
        self._check(self._L.vdl_comm_init(self._c, int(rank), int(world), ctypes.c_char_p(unique_id)))

    def comm_init_host(self, rank, world, all_gather, all_to_all):
        """Collectives supplied by the caller over host memory:
        all_gather(send: bytes) -> list of `world` bytes objects (rank order);
        all_to_all(pieces: list of `world` bytes objects, one per destination) -> list of `world` bytes objects, one per source."""
        def c_all_gather(_user, send, recv, nbytes):
            try:
                parts = all_gather(ctypes.string_at(send, nbytes))
                for r, part in enumerate(parts):
                    if len(part) != nbytes:
                        return 1
                    ctypes.memmove(recv + r * nbytes, part, nbytes)
                return 0
            except Exception:              # noqa: BLE001 -- must not propagate through the C frame
                import traceback
                traceback.print_exc()
                return 1

        def c_all_to_all(_user, send, send_bytes, recv, recv_bytes):
            try:
                pieces, at = [], 0
                for r in range(world):
                    pieces.append(ctypes.string_at(send + at, send_bytes[r]) if send_bytes[r] else b"")
                    at += send_bytes[r]
                got = all_to_all(pieces)
                at = 0
                for r in range(world):
                    if len(got[r]) != recv_bytes[r]:
                        return 1
                    if recv_bytes[r]:
                        ctypes.memmove(recv + at, got[r], recv_bytes[r])
                    at += recv_bytes[r]
                return 0
            except Exception:              # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        self._comm_host = _lib.CommHost(None, _lib.ALL_GATHER_FN(c_all_gather), _lib.ALL_TO_ALL_FN(c_all_to_all))     # kept alive with the engine
        self._check(self._L.vdl_comm_init_host(self._c, int(rank), int(world), ctypes.byref(self._comm_host)))

    def comm_info(self):
        rank, world, name = ctypes.c_int(), ctypes.c_int(), ctypes.c_char_p()
        self._check(self._L.vdl_comm_info(self._c, ctypes.byref(rank), ctypes.byref(world), ctypes.byref(name)))
        return rank.value, world.value, name.value.decode()

    # ---- catalog ----
    def register_tensor(self, name, tensor):
        """Borrow a 1-D integer torch tensor that already lives in HBM."""
        if not tensor.is_cuda or tensor.dim() != 1 or not tensor.is_contiguous():
            raise VdlError(_lib.VDL_ERR_ARG, "register_tensor needs a contiguous 1-D device tensor")
        self._keep[name] = tensor
        self._check(self._L.vdl_register_column(self._c, name.encode(), ctypes.c_void_p(tensor.data_ptr()),
                                                tensor.element_size(), tensor.numel()))

    def register_pointer(self, name, dev_ptr, elem_bytes, nrows):
        """Borrow nrows integers of elem_bytes each at a device address the caller keeps alive (vdl_register_column)."""
        self._check(self._L.vdl_register_column(self._c, name.encode(), ctypes.c_void_p(dev_ptr), elem_bytes, nrows))

    def upload(self, name, array):
        a = np.ascontiguousarray(array)
        if a.dtype.kind != "i":
            raise VdlError(_lib.VDL_ERR_ARG, "columns are signed integers")
        self._check(self._L.vdl_upload_column(self._c, name.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                              a.dtype.itemsize, a.shape[0]))

    def generate(self, spec, row0, nrows, seed=datagen.SEED):
        """Materialise rows [row0, row0+nrows) of a synthetic column directly in HBM."""
        self._check(self._L.vdl_generate_column(self._c, spec.name.encode(), np.dtype(spec.dtype).itemsize, row0, nrows,
                                                seed, spec.lo, spec.hi, spec.mul, spec.add))

    def download(self, name):
        w, n = ctypes.c_int(), ctypes.c_int64()
        self._check(self._L.vdl_column_info(self._c, name.encode(), ctypes.byref(w), ctypes.byref(n), None))
        out = np.empty(n.value, dtype={1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[w.value])
        self._check(self._L.vdl_download_column(self._c, name.encode(), out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        return out

    def column_device(self, name):
        """A registered / generated column as it lies in HBM: `DeviceValues`-like object with `__cuda_array_interface__`
        (`torch.as_tensor(x, device=...)` wraps it without a copy; checkers use it to read the very bytes the engine scans)."""
        w, n, ptr = ctypes.c_int(), ctypes.c_int64(), ctypes.c_void_p()
        self._check(self._L.vdl_column_info(self._c, name.encode(), ctypes.byref(w), ctypes.byref(n), ctypes.byref(ptr)))

        class _Column:
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "<i%d" % w.value, "data": (ptr.value or 0, False), "version": 3, "strides": None}
            owner = self
        return _Column()

    def drop(self, name):
        self._keep.pop(name, None)
        self._check(self._L.vdl_drop_column(self._c, name.encode()))

    # ---- programs ----
    def parse(self, vdl_text):
        data = vdl_text.encode() if isinstance(vdl_text, str) else vdl_text
        h = ctypes.c_void_p()
        rc = self._L.vdl_parse(self._c, data, len(data), ctypes.byref(h))
        self._check(rc)
        plan = Plan(self, h, vdl_text)
        self._plans.add(plan)
        return plan

    def run_vdl(self, vdl_text, fuse=True):
        plan = self.parse(vdl_text)
        try:
            plan.set_fusion(fuse)
            return plan.run()
        finally:
            plan.close()
