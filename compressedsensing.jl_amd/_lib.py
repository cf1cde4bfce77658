"""ctypes binding of csrc/libcsmp.so -- the C ABI declared in include/csmp.h.

This is the Python twin of the `ccall` stubs in julia/CompressedSensingAMD.jl.  There is NO CPU
fallback: if the shared library is missing, or no MI355X is visible, every compute entry point
raises `CsmpError` loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libcsmp.so")

OK, EINVAL, EDIM, ERANGE, EHIP, ESTATE, ENOMEM, ERCCL, EIO = 0, -1, -2, -3, -4, -5, -6, -7, -8
F32, F64 = 0, 1
HOST, DEVICE, HOST_STREAMED = 0, 1, 2
ALGO_MP, ALGO_OMP, ALGO_GOMP, ALGO_FR, ALGO_SP, ALGO_OMPR = 0, 1, 2, 3, 4, 5
STOP_EPS, STOP_STAG, STOP_FULL = 1, 2, 4
# csmp_set_option keys (include/csmp.h)
OPT_BATCH_CERT, OPT_BATCH_GRAM, OPT_BATCH_WINDOW, OPT_PIPELINE, OPT_SOLVES_IN_FLIGHT, OPT_SCREENED_SWEEP, OPT_BATCH_SCREEN = 1, 2, 3, 4, 9, 10, 11
OPTIONS = {"batch_cert": OPT_BATCH_CERT, "batch_gram": OPT_BATCH_GRAM, "batch_window": OPT_BATCH_WINDOW, "pipeline": OPT_PIPELINE,
           "solves_in_flight": OPT_SOLVES_IN_FLIGHT, "screened_sweep": OPT_SCREENED_SWEEP, "batch_screen": OPT_BATCH_SCREEN}

i64 = C.c_int64
vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/csmp.h declares
SIGNATURES = {
    "csmp_version": (C.c_int, []),
    "csmp_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "csmp_destroy": (C.c_int, [vp]),
    "csmp_last_error": (C.c_char_p, [vp]),
    "csmp_set_stream": (C.c_int, [vp, vp]),
    "csmp_sync": (C.c_int, [vp]),
    "csmp_device_info": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(i64)]),
    "csmp_set_dictionary": (C.c_int, [vp, vp, i64, i64, i64, C.c_int, C.c_int]),
    "csmp_dictionary_file_write": (C.c_int, [C.c_char_p, vp, i64, i64, i64, C.c_int]),
    "csmp_dictionary_file_info": (C.c_int, [C.c_char_p, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_int)]),
    "csmp_set_dictionary_file": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "csmp_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "csmp_shard_config": (C.c_int, [vp, i64]),
    "csmp_shard_record_bytes": (i64, [vp]),
    "csmp_shard_sweep": (C.c_int, [vp, C.c_double, C.c_int, vp]),
    "csmp_shard_append": (C.c_int, [vp, vp, C.c_int]),
    "csmp_shard_range": (C.c_int, [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]),
    "csmp_comm_id": (C.c_int, [vp]),
    "csmp_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "csmp_comm_free": (C.c_int, [vp]),
    "csmp_omp_sharded": (C.c_int, [vp, vp, C.c_int, i64, i64, C.c_int, i64, C.c_double, C.c_int, vp, vp, vp, C.c_int]),
    "csmp_pack_block_device": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "csmp_unpack_gathered_device": (C.c_int, [vp, vp, i64, i64, C.c_int, vp, vp, vp]),
    "csmp_pack_results": (C.c_int, [vp, vp, vp, i64, i64, vp]),
    "csmp_unpack_results": (C.c_int, [vp, i64, i64, vp, vp, vp]),
    "csmp_mp": (C.c_int, [vp, vp, C.c_int, i64, vp, vp, i64, vp, vp, C.POINTER(i64)]),
    "csmp_omp": (C.c_int, [vp, vp, C.c_int, i64, C.c_double, vp, vp, C.POINTER(i64), vp]),
    "csmp_gomp": (C.c_int, [vp, vp, C.c_int, i64, i64, C.c_double, vp, vp, C.POINTER(i64), vp]),
    "csmp_sp": (C.c_int, [vp, vp, C.c_int, i64, C.c_double, i64, vp, vp, C.POINTER(i64), C.POINTER(i64)]),
    "csmp_fr": (C.c_int, [vp, vp, C.c_int, i64, C.c_double, C.c_double, vp, vp, C.POINTER(i64), vp]),
    "csmp_fr_scores": (C.c_int, [vp, vp]),
    "csmp_srr": (C.c_int, [vp, vp, C.c_int, i64, C.c_double, i64, C.c_int, i64, vp, vp, C.POINTER(i64), C.POINTER(i64)]),
    "csmp_srr_from": (C.c_int, [vp, vp, C.c_int, i64, C.c_double, i64, vp, i64, vp, vp, C.POINTER(i64), C.POINTER(i64)]),
    "csmp_rmp_delta": (C.c_int, [vp, vp, C.c_int, C.c_double, i64, i64, vp, vp, C.POINTER(i64)]),
    "csmp_rmp_k": (C.c_int, [vp, vp, C.c_int, i64, i64, vp, vp, C.POINTER(i64)]),
    "csmp_foba": (C.c_int, [vp, vp, C.c_int, C.c_double, i64, vp, vp, C.POINTER(i64)]),
    "csmp_br": (C.c_int, [vp, vp, C.c_int, C.c_double, C.c_double, i64, C.c_int, vp, vp, C.POINTER(i64)]),
    "csmp_fr_batch": (C.c_int, [vp, vp, C.c_int, i64, i64, C.c_int, i64, C.c_double, C.c_double, vp, vp, vp, C.c_int]),
    "csmp_ompr": (C.c_int, [vp, vp, C.c_int, i64, C.c_double, i64, vp, vp, C.POINTER(i64), C.POINTER(i64)]),
    "csmp_omp_batch": (C.c_int, [vp, vp, C.c_int, i64, i64, C.c_int, i64, C.c_double, vp, vp, vp, C.c_int]),
    "csmp_gomp_batch": (C.c_int, [vp, vp, C.c_int, i64, i64, C.c_int, i64, i64, C.c_double, vp, vp, vp, C.c_int]),
    "csmp_sp_batch": (C.c_int, [vp, vp, C.c_int, i64, i64, i64, C.c_double, i64, vp, vp, vp, vp]),
    "csmp_omp_batch_mfma": (C.c_int, [vp, vp, C.c_int, i64, i64, C.c_int, i64, C.c_double, vp, vp, vp, C.c_int]),
    "csmp_batch_stats": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                   C.POINTER(C.c_double)]),
    "csmp_screened_stats": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64), C.c_int]),
    "csmp_set_option": (C.c_int, [vp, C.c_int, i64]),
    "csmp_get_option": (C.c_int, [vp, C.c_int, C.POINTER(i64)]),
    "csmp_solver_begin": (C.c_int, [vp, C.c_int, vp, C.c_int, i64, vp, vp, i64]),
    "csmp_solver_step": (C.c_int, [vp, i64]),
    "csmp_solver_acquire": (C.c_int, [vp, i64]),
    "csmp_solver_remove": (C.c_int, [vp, i64]),
    "csmp_solver_state": (C.c_int, [vp, vp, vp, C.POINTER(i64), C.POINTER(C.c_double), vp, C.POINTER(C.c_int)]),
    "csmp_sweep": (C.c_int, [vp, vp, vp, i64, vp, vp]),
    "csmp_lstsq": (C.c_int, [vp, vp, i64, vp, C.c_int, vp]),
}


# measurement and test hooks: include/csmp_internal.h (not part of the drop-in boundary)
INTERNAL_SIGNATURES = {
    "csmp_profile_enable": (C.c_int, [vp, C.c_int]),
    "csmp_profile_read": (C.c_int, [vp, C.POINTER(i64), C.POINTER(C.c_double), C.c_int]),
    "csmp_bench_sweep": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "csmp_profile_overhead": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
    "csmp_live_resources": (C.c_int, [C.POINTER(i64)] * 6),
    "csmp_profile_window": (C.c_int, [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "csmp_sweep_config": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "csmp_tune": (C.c_int, [vp, C.c_int, i64]),
    "csmp_batch_layout": (C.c_int, [vp, C.POINTER(i64), C.POINTER(C.c_int)]),
    "csmp_batch_screen_kernel": (C.c_char_p, [vp]),
}
TUNE = {"sweep_grid": 2, "sweep_unit": 3, "tick_grid": 4, "batch_budget_mib": 5, "diag_split": 6, "swap_refuse": 7, "rebuild_direct": 8, "sweep_dyn": 9, "tick_order": 10, "claim_pools": 11, "pipelines": 12, "pair_lds_kib": 13, "pair_split": 14, "sweep_lds_kib": 15, "sweep_short": 16, "phase_rows": 17, "fail_alloc": 18, "screen_static": 19}  # CSMP_TUNE_* (include/csmp_internal.h)

COMM_ID_BYTES = 128  # CSMP_COMM_ID_BYTES


def live_resources():
    """what the library holds right now, process-wide (csmp_internal.h): all zero when no context exists"""
    v = [i64(0) for _ in range(6)]
    rc = lib().csmp_live_resources(*[C.byref(x) for x in v])
    if rc != OK:
        raise CsmpError(rc, "csmp_live_resources")
    return dict(zip(("device_bytes", "device_blocks", "pinned_bytes", "registered_ranges", "events", "streams"), (int(x.value) for x in v)))


def comm_id():
    """rank 0: a fresh RCCL communicator id (bytes) to hand to every rank's Context.comm_init."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().csmp_comm_id(C.cast(buf, vp))
    if rc != 0:
        raise CsmpError(rc, lib().csmp_last_error(None).decode())
    return buf.raw


def write_dictionary_file(path, A):
    """A (numpy, M x N, float32 / float64) -> a dictionary file (include/csmp.h: 64-byte header + the columns padded to 16 bytes).
    Host only: no GPU is touched."""
    A = np.asarray(A)
    if A.ndim != 2 or A.dtype not in (np.float32, np.float64):
        raise ValueError("A must be a float32 or float64 matrix")
    if not A.flags.f_contiguous:
        A = np.asfortranarray(A)
    M, N = A.shape
    rc = lib().csmp_dictionary_file_write(os.fsencode(path), ptr(A), i64(M), i64(N), i64(M), dtype_code(A.dtype))
    if rc != 0:
        raise CsmpError(rc, f"cannot write {path}")


def dictionary_file_info(path):
    """(M, N, numpy dtype) of a dictionary file."""
    M, N, dt = i64(0), i64(0), C.c_int(0)
    rc = lib().csmp_dictionary_file_info(os.fsencode(path), C.byref(M), C.byref(N), C.byref(dt))
    if rc != 0:
        raise CsmpError(rc, f"{path} is not a dictionary file")
    return int(M.value), int(N.value), (np.float32 if dt.value == dtype_code(np.dtype(np.float32)) else np.float64)


class CsmpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcsmp error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libcsmp.so (built by `make -C csrc` / __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CsmpError(ESTATE, f"{LIB_PATH} is missing -- build it with `make -C {os.path.dirname(LIB_PATH)}`; "
                                    "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(INTERNAL_SIGNATURES.items()):
            f = getattr(L, name)  # AttributeError if the ABI and the header disagree
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(vp)


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError(f"unsupported element type {dt}: the dictionary and b must be float32 or float64")


class Context:
    """One GPU + one HIP stream + the resident dictionary (csmp_ctx)."""
    last_status = OK  # status of the most recent call

    def __init__(self, device=0):
        self._h = vp()
        L = lib()
        rc = L.csmp_create(C.byref(self._h), int(device))
        if rc != OK:
            msg = L.csmp_last_error(None).decode()
            self._h = None
            raise CsmpError(rc, msg)
        self.M = self.N = 0
        self.dtype = None
        self._keep = None  # keeps a borrowed device tensor alive

    def close(self):
        if getattr(self, "_h", None):
            lib().csmp_destroy(self._h)
            self._h = None
        self._parent = None

    def clone(self):
        """A second context on the same GPU borrowing this one's resident dictionary (csmp_clone): what every
        step-level functor works on, so that two of them never share solver state."""
        c = Context.__new__(Context)
        c._h = vp()
        c.M, c.N, c.dtype, c._keep = self.M, self.N, self.dtype, self._keep
        c._parent = self  # the borrowed dictionary must outlive the clone
        rc = lib().csmp_clone(self._h, C.byref(c._h))
        if rc != OK:
            c._h = None
            self.check(rc)
        return c

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        self.last_status = rc
        if rc < OK:
            raise CsmpError(rc, lib().csmp_last_error(self._h).decode())

    def call(self, name, *args):
        self.check(getattr(lib(), name)(self._h, *args))

    # ---- dictionary
    def set_dictionary(self, A, streamed=False):
        """A: numpy array (host; copied once to HBM) or a torch CUDA tensor holding the
        column-major dictionary as a (N, M) row-major tensor, i.e. `A_torch[j]` is atom j.
        streamed=True (numpy only): A stays in host memory, mapped into the device; every sweep crosses the host link
        (CSMP_HOST_STREAMED: for a dictionary larger than HBM).  The array is kept alive by this object."""
        if isinstance(A, np.ndarray):
            if A.ndim != 2:
                raise ValueError("A must be a matrix")
            if not A.flags.f_contiguous:
                A = np.asfortranarray(A)
            M, N = A.shape
            self.call("csmp_set_dictionary", ptr(A), i64(M), i64(N), i64(M), dtype_code(A.dtype), HOST_STREAMED if streamed else HOST)
            self.dtype = A.dtype
            self._keep = A if streamed else None
        else:  # torch tensor on the GPU, shape (N, M): rows are atoms
            import torch
            if not (isinstance(A, torch.Tensor) and A.is_cuda and A.dim() == 2 and A.is_contiguous()):
                raise ValueError("device dictionary must be a contiguous 2-D CUDA tensor of shape (N atoms, M rows)")
            N, M = A.shape
            dt = np.float32 if A.dtype == torch.float32 else np.float64 if A.dtype == torch.float64 else None
            if dt is None:
                raise TypeError("device dictionary must be float32 or float64")
            self.call("csmp_set_dictionary", vp(A.data_ptr()), i64(M), i64(N), i64(M), dtype_code(dt), DEVICE)
            self.dtype = np.dtype(dt)
            self._keep = A
        self.M, self.N = int(M), int(N)

    def set_dictionary_file(self, path, streamed=False):
        """A dictionary file (write_dictionary_file) read to where it will live: HBM, or mapped host memory (streamed=True)."""
        M, N, dt = dictionary_file_info(path)
        self.call("csmp_set_dictionary_file", os.fsencode(path), HOST_STREAMED if streamed else DEVICE)
        self.dtype = np.dtype(dt)
        self._keep = None
        self.M, self.N = M, N

    def sync(self):
        self.call("csmp_sync")

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        mem = i64(0)
        self.call("csmp_device_info", name, 256, C.byref(cus), C.byref(mem))
        return name.value.decode(), cus.value, mem.value

    def _b(self, b):
        b = np.ascontiguousarray(b)
        if b.dtype not in (np.float32, np.float64):
            b = b.astype(np.float64)
        if b.shape != (self.M,):
            raise CsmpError(EDIM, f"length(b) = {b.shape} but size(A, 1) = {self.M}")
        return b

    # ---- drivers
    def omp(self, b, k, eps):
        b = self._b(b)
        cap = max(int(k), 1)
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        order = np.zeros(cap, np.int64)
        nnz = i64(0)
        self.call("csmp_omp", ptr(b), dtype_code(b.dtype), i64(int(k)), C.c_double(eps), ptr(idx), ptr(val),
                  C.byref(nnz), ptr(order))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), order[:n].copy()

    def gomp(self, b, l, k, eps):
        b = self._b(b)
        cap = max(int(k) + int(l), 1)
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        order = np.zeros(cap, np.int64)
        nnz = i64(0)
        self.call("csmp_gomp", ptr(b), dtype_code(b.dtype), i64(int(l)), i64(int(k)), C.c_double(eps), ptr(idx),
                  ptr(val), C.byref(nnz), ptr(order))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), order[:n].copy()

    def mp(self, b, k, idx0=None, val0=None):
        b = self._b(b)
        idx0 = np.zeros(0, np.int64) if idx0 is None else np.ascontiguousarray(idx0, dtype=np.int64)
        val0 = np.zeros(0, np.float64) if val0 is None else np.ascontiguousarray(val0, dtype=np.float64)
        cap = max(int(k) + len(idx0), 1)
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        nnz = i64(0)
        self.call("csmp_mp", ptr(b), dtype_code(b.dtype), i64(int(k)), ptr(idx0), ptr(val0), i64(len(idx0)),
                  ptr(idx), ptr(val), C.byref(nnz))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy()

    def sp(self, b, k, delta, maxiter=-1):
        b = self._b(b)
        cap = max(2 * int(k), 1)
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        nnz = i64(0)
        iters = i64(0)
        self.call("csmp_sp", ptr(b), dtype_code(b.dtype), i64(int(k)), C.c_double(delta), i64(int(maxiter)),
                  ptr(idx), ptr(val), C.byref(nnz), C.byref(iters))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), iters.value

    def ompr(self, b, k, delta, maxiter=-1):
        b = self._b(b)
        idx = np.zeros(max(int(k), 1), np.int64)
        val = np.zeros(max(int(k), 1), np.float64)
        nnz = i64(0)
        iters = i64(0)
        self.call("csmp_ompr", ptr(b), dtype_code(b.dtype), i64(int(k)), C.c_double(delta), i64(int(maxiter)),
                  ptr(idx), ptr(val), C.byref(nnz), C.byref(iters))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), iters.value

    def fr(self, b, k, max_eps=0.0, min_delta=0.0):
        b = self._b(b)
        cap = max(int(k), 1)
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        order = np.zeros(cap, np.int64)
        nnz = i64(0)
        self.call("csmp_fr", ptr(b), dtype_code(b.dtype), i64(int(k)), C.c_double(max_eps), C.c_double(min_delta),
                  ptr(idx), ptr(val), C.byref(nnz), ptr(order))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), order[:n].copy()

    def srr(self, b, k, delta=1e-12, maxiter=-1, initialization=1, l=1, init=None):
        b = self._b(b)
        cap = int(k) + int(l) + 1
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        nnz = i64(0)
        iters = i64(0)
        if int(initialization) == 3 and init is not None:  # (without a draw: csmp_srr refuses initialization 3 itself)
            init = np.ascontiguousarray(init, np.int64)
            if init.size != int(k):
                raise ValueError("srr: initialization = 3 needs k initial atoms")
            self.call("csmp_srr_from", ptr(b), dtype_code(b.dtype), i64(int(k)), C.c_double(delta), i64(int(maxiter)),
                      ptr(init), i64(int(l)), ptr(idx), ptr(val), C.byref(nnz), C.byref(iters))
            n = nnz.value
            return idx[:n].copy(), val[:n].copy(), iters.value
        self.call("csmp_srr", ptr(b), dtype_code(b.dtype), i64(int(k)), C.c_double(delta), i64(int(maxiter)),
                  int(initialization), i64(int(l)), ptr(idx), ptr(val), C.byref(nnz), C.byref(iters))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), iters.value

    def _stepwise_cap(self, kmax):
        lim = min(self.M, self.N, 4095)
        return lim if kmax is None or kmax <= 0 else min(int(kmax), lim)

    def rmp(self, b, delta_or_k, maxiter=1, kmax=None):
        """rmp(A,b,δ,maxiter) for a float second argument, rmp(A,b,k) for an int."""
        b = self._b(b)
        cap = self._stepwise_cap(kmax)
        idx = np.zeros(cap + 1, np.int64)
        val = np.zeros(cap + 1, np.float64)
        nnz = i64(0)
        if isinstance(delta_or_k, (int, np.integer)):
            self.call("csmp_rmp_k", ptr(b), dtype_code(b.dtype), i64(int(delta_or_k)), i64(cap), ptr(idx), ptr(val), C.byref(nnz))
        else:
            self.call("csmp_rmp_delta", ptr(b), dtype_code(b.dtype), C.c_double(float(delta_or_k)), i64(int(maxiter)), i64(cap),
                      ptr(idx), ptr(val), C.byref(nnz))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy()

    def foba(self, b, delta, kmax=None):
        b = self._b(b)
        cap = self._stepwise_cap(kmax)
        idx = np.zeros(cap + 1, np.int64)
        val = np.zeros(cap + 1, np.float64)
        nnz = i64(0)
        self.call("csmp_foba", ptr(b), dtype_code(b.dtype), C.c_double(float(delta)), i64(cap), ptr(idx), ptr(val), C.byref(nnz))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy()

    def br(self, b, max_eps=float("inf"), max_delta=float("inf"), k=0, lace=False):
        b = self._b(b)
        idx = np.zeros(self.N + 1, np.int64)
        val = np.zeros(self.N + 1, np.float64)
        nnz = i64(0)
        self.call("csmp_br", ptr(b), dtype_code(b.dtype), C.c_double(float(max_eps)), C.c_double(float(max_delta)),
                  i64(int(k)), int(bool(lace)), ptr(idx), ptr(val), C.byref(nnz))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy()

    def fr_scores(self):
        d2 = np.zeros(self.N, np.float64)
        self.call("csmp_fr_scores", ptr(d2))
        return d2

    def omp_batch(self, B, k, eps):
        """Host matrix B (M x nsig, column-major) -> (idx k x nsig, val, nnz) numpy arrays."""
        B = np.asfortranarray(B)
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        M, nsig = B.shape
        if M != self.M:
            raise CsmpError(EDIM, f"size(B, 1) = {M} but size(A, 1) = {self.M}")
        idx = np.zeros((int(k), nsig), np.int64, order="F")
        val = np.zeros((int(k), nsig), np.float64, order="F")
        nnz = np.zeros(nsig, np.int64)
        self.call("csmp_omp_batch", ptr(B), dtype_code(B.dtype), i64(M), i64(nsig), HOST, i64(int(k)),
                  C.c_double(eps), ptr(idx), ptr(val), ptr(nnz), HOST)
        return idx, val, nnz

    def gomp_batch(self, B, l, k, eps):
        """Host matrix B (M x nsig, column-major) -> (idx k x nsig, val, nnz): gomp for every column, two solves in flight."""
        B = np.asfortranarray(B)
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        M, nsig = B.shape
        if M != self.M:
            raise CsmpError(EDIM, f"size(B, 1) = {M} but size(A, 1) = {self.M}")
        idx = np.zeros((int(k), nsig), np.int64, order="F")
        val = np.zeros((int(k), nsig), np.float64, order="F")
        nnz = np.zeros(nsig, np.int64)
        self.call("csmp_gomp_batch", ptr(B), dtype_code(B.dtype), i64(M), i64(nsig), HOST, i64(int(l)), i64(int(k)),
                  C.c_double(eps), ptr(idx), ptr(val), ptr(nnz), HOST)
        return idx, val, nnz

    def gomp_batch_device(self, B, l, k, eps, idx, val, nnz):
        """torch CUDA tensors: B (nsig, M) rows = signals; outputs idx (nsig, k) int64, val (nsig, k) float64, nnz (nsig,) int64."""
        import torch
        nsig, M = B.shape
        assert B.is_cuda and B.is_contiguous() and M == self.M
        assert idx.dtype == torch.int64 and val.dtype == torch.float64 and nnz.dtype == torch.int64
        assert idx.is_contiguous() and val.is_contiguous() and idx.shape == (nsig, int(k)) and val.shape == (nsig, int(k))
        code = F32 if B.dtype == torch.float32 else F64
        self.call("csmp_gomp_batch", vp(B.data_ptr()), code, i64(M), i64(nsig), DEVICE, i64(int(l)), i64(int(k)), C.c_double(eps),
                  vp(idx.data_ptr()), vp(val.data_ptr()), vp(nnz.data_ptr()), DEVICE)

    def sp_batch(self, B, k, delta=1e-12, maxiter=-1):
        """Host matrix B (M x nsig, column-major) -> (idx k x nsig, val, nnz, iters): sp for every column, several solves in flight."""
        B = np.asfortranarray(B)
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        M, nsig = B.shape
        if M != self.M:
            raise CsmpError(EDIM, f"size(B, 1) = {M} but size(A, 1) = {self.M}")
        idx = np.zeros((int(k), nsig), np.int64, order="F")
        val = np.zeros((int(k), nsig), np.float64, order="F")
        nnz = np.zeros(nsig, np.int64)
        its = np.zeros(nsig, np.int64)
        self.call("csmp_sp_batch", ptr(B), dtype_code(B.dtype), i64(M), i64(nsig), i64(int(k)), C.c_double(delta), i64(int(maxiter)),
                  ptr(idx), ptr(val), ptr(nnz), ptr(its))
        return idx, val, nnz, its

    def fr_batch(self, B, k, max_eps=0.0, min_delta=0.0):
        """fr for every column of the host matrix B (M x nsig, column-major) -> (idx k x nsig, val, nnz)."""
        B = np.asfortranarray(B)
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        M, nsig = B.shape
        if M != self.M:
            raise CsmpError(EDIM, f"size(B, 1) = {M} but size(A, 1) = {self.M}")
        idx = np.zeros((int(k), nsig), np.int64, order="F")
        val = np.zeros((int(k), nsig), np.float64, order="F")
        nnz = np.zeros(nsig, np.int64)
        self.call("csmp_fr_batch", ptr(B), dtype_code(B.dtype), i64(M), i64(nsig), HOST, i64(int(k)),
                  C.c_double(max_eps), C.c_double(min_delta), ptr(idx), ptr(val), ptr(nnz), HOST)
        return idx, val, nnz

    def fr_batch_device(self, B, k, max_eps, min_delta, idx, val, nnz):
        """torch CUDA tensors as in omp_batch_device.  Only enqueues work; call sync()."""
        import torch
        nsig, M = B.shape
        assert B.is_cuda and B.is_contiguous() and M == self.M
        assert idx.dtype == torch.int64 and val.dtype == torch.float64 and nnz.dtype == torch.int64
        assert idx.is_contiguous() and val.is_contiguous() and idx.shape == (nsig, int(k)) and val.shape == (nsig, int(k))
        code = F32 if B.dtype == torch.float32 else F64
        self.call("csmp_fr_batch", vp(B.data_ptr()), code, i64(M), i64(nsig), DEVICE, i64(int(k)), C.c_double(max_eps),
                  C.c_double(min_delta), vp(idx.data_ptr()), vp(val.data_ptr()), vp(nnz.data_ptr()), DEVICE)

    def omp_batch_mfma(self, B, k, eps):
        """Batched (MFMA-screened) variant of omp_batch: same inputs and outputs."""
        B = np.asfortranarray(B)
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        M, nsig = B.shape
        if M != self.M:
            raise CsmpError(EDIM, f"size(B, 1) = {M} but size(A, 1) = {self.M}")
        idx = np.zeros((int(k), nsig), np.int64, order="F")
        val = np.zeros((int(k), nsig), np.float64, order="F")
        nnz = np.zeros(nsig, np.int64)
        self.call("csmp_omp_batch_mfma", ptr(B), dtype_code(B.dtype), i64(M), i64(nsig), HOST, i64(int(k)),
                  C.c_double(eps), ptr(idx), ptr(val), ptr(nnz), HOST)
        return idx, val, nnz

    def omp_batch_mfma_device(self, B, k, eps, idx, val, nnz):
        """torch CUDA tensors as in omp_batch_device; synchronises once at the end."""
        import torch
        nsig, M = B.shape
        assert B.is_cuda and B.is_contiguous() and M == self.M
        assert idx.dtype == torch.int64 and val.dtype == torch.float64 and nnz.dtype == torch.int64
        assert idx.is_contiguous() and val.is_contiguous() and idx.shape == (nsig, int(k)) and val.shape == (nsig, int(k))
        code = F32 if B.dtype == torch.float32 else F64
        self.call("csmp_omp_batch_mfma", vp(B.data_ptr()), code, i64(M), i64(nsig), DEVICE, i64(int(k)), C.c_double(eps),
                  vp(idx.data_ptr()), vp(val.data_ptr()), vp(nnz.data_ptr()), DEVICE)

    def batch_screen_kernel(self):
        return lib().csmp_batch_screen_kernel(self._h).decode()

    # ---- signals sharded over GPUs with the collective inside the library (csmp_comm_*, csmp_omp_sharded)
    def comm_init(self, comm_id, rank, world):
        """ncclCommInitRank on this context's GPU; comm_id: the COMM_ID_BYTES bytes rank 0 got from comm_id()."""
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self.call("csmp_comm_init", C.cast(buf, vp), int(rank), int(world))

    def comm_free(self):
        self.call("csmp_comm_free")

    def pack_block_device(self, idx, val, nnz, rows):
        """torch CUDA tensors idx / val (nloc, k), nnz (nloc,) -> (rows, 2k+1) float64 tensor on the device (rows >= nloc, surplus zero)."""
        import torch
        nloc, k = idx.shape
        out = torch.empty((int(rows), 2 * k + 1), dtype=torch.float64, device=idx.device)
        self.call("csmp_pack_block_device", vp(idx.data_ptr() if nloc else 0), vp(val.data_ptr() if nloc else 0), vp(nnz.data_ptr() if nloc else 0),
                  i64(k), i64(nloc), i64(int(rows)), vp(out.data_ptr()))
        return out

    def unpack_gathered_device(self, gathered, k, nsig, world):
        """(world * rows, 2k+1) float64 CUDA tensor of every rank's packed block -> idx (nsig, k) int64, val (nsig, k), nnz (nsig,) on the device."""
        import torch
        dev = gathered.device
        idx = torch.empty((int(nsig), int(k)), dtype=torch.int64, device=dev)
        val = torch.empty((int(nsig), int(k)), dtype=torch.float64, device=dev)
        nnz = torch.empty(int(nsig), dtype=torch.int64, device=dev)
        self.call("csmp_unpack_gathered_device", vp(gathered.data_ptr()), i64(int(k)), i64(int(nsig)), int(world), vp(idx.data_ptr()),
                  vp(val.data_ptr()), vp(nnz.data_ptr()))
        return idx, val, nnz

    def omp_sharded(self, B_local, nsig, k, eps, method="exact"):
        """This rank's block B_local (M x nloc, host) of `nsig` signals in all -> (idx k x nsig, val, nnz) of ALL signals."""
        B = np.asfortranarray(B_local)
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        if B.shape[0] != self.M:
            raise CsmpError(EDIM, f"size(B, 1) = {B.shape[0]} but size(A, 1) = {self.M}")
        idx = np.zeros((int(k), int(nsig)), np.int64, order="F")
        val = np.zeros((int(k), int(nsig)), np.float64, order="F")
        nnz = np.zeros(int(nsig), np.int64)
        self.call("csmp_omp_sharded", ptr(B), dtype_code(B.dtype), i64(self.M), i64(int(nsig)), HOST, i64(int(k)), C.c_double(eps),
                  {"exact": 0, "mfma": 1}[method], ptr(idx), ptr(val), ptr(nnz), HOST)
        return idx, val, nnz

    def omp_sharded_device(self, B_local, nsig, k, eps, idx, val, nnz, method="exact"):
        """torch CUDA tensors: B_local (nloc, M) rows = this rank's signals; idx / val (nsig, k), nnz (nsig): all signals."""
        import torch
        nloc, M = B_local.shape
        assert B_local.is_cuda and B_local.is_contiguous() and M == self.M
        assert idx.dtype == torch.int64 and val.dtype == torch.float64 and nnz.dtype == torch.int64
        assert idx.is_contiguous() and val.is_contiguous() and idx.shape == (int(nsig), int(k)) and val.shape == (int(nsig), int(k))
        code = F32 if B_local.dtype == torch.float32 else F64
        self.call("csmp_omp_sharded", vp(B_local.data_ptr() if nloc else 0), code, i64(M), i64(int(nsig)), DEVICE, i64(int(k)), C.c_double(eps),
                  {"exact": 0, "mfma": 1}[method], vp(idx.data_ptr()), vp(val.data_ptr()), vp(nnz.data_ptr()), DEVICE)

    def batch_layout(self):
        n, st = i64(0), C.c_int(0)
        self.call("csmp_batch_layout", C.byref(n), C.byref(st))
        return {"screen_signals": n.value, "streams": st.value}

    def batch_stats(self):
        v = [i64(0) for _ in range(5)]
        ms = C.c_double(0)
        self.call("csmp_batch_stats", *[C.byref(x) for x in v], C.byref(ms))
        return {"signals": v[0].value, "resolved_exactly": v[1].value, "uncertain": v[2].value, "illcond": v[3].value,
                "screen_launches": v[4].value, "screen_ms": ms.value}

    def screened_stats(self, reset=False):
        """Screened solves (option screened_sweep) made by this context, and how many were repeated with the exact sweep."""
        a, b = i64(0), i64(0)
        self.call("csmp_screened_stats", C.byref(a), C.byref(b), int(bool(reset)))
        return {"solves": a.value, "fallbacks": b.value}

    def omp_batch_device(self, B, k, eps, idx, val, nnz):
        """torch CUDA tensors: B (nsig, M) rows = signals; outputs idx (nsig, k) int64,
        val (nsig, k) float64, nnz (nsig,) int64.  Only enqueues work; call sync()."""
        import torch
        nsig, M = B.shape
        assert B.is_cuda and B.is_contiguous() and M == self.M
        assert idx.dtype == torch.int64 and val.dtype == torch.float64 and nnz.dtype == torch.int64
        assert idx.is_contiguous() and val.is_contiguous() and idx.shape == (nsig, int(k)) and val.shape == (nsig, int(k))
        code = F32 if B.dtype == torch.float32 else F64
        self.call("csmp_omp_batch", vp(B.data_ptr()), code, i64(M), i64(nsig), DEVICE, i64(int(k)), C.c_double(eps),
                  vp(idx.data_ptr()), vp(val.data_ptr()), vp(nnz.data_ptr()), DEVICE)

    # ---- step level
    def solver_begin(self, algo, b, kcap, idx0=None, val0=None):
        b = self._b(b)
        idx0 = np.zeros(0, np.int64) if idx0 is None else np.ascontiguousarray(idx0, dtype=np.int64)
        val0 = np.zeros(0, np.float64) if val0 is None else np.ascontiguousarray(val0, dtype=np.float64)
        self.call("csmp_solver_begin", int(algo), ptr(b), dtype_code(b.dtype), i64(int(kcap)), ptr(idx0), ptr(val0),
                  i64(len(idx0)))

    def solver_step(self, l=1):
        self.call("csmp_solver_step", i64(int(l)))

    def solver_acquire(self, k):
        """sp_acquisition!(P, x, k) (SP) / oblivious_acquisition!(P, x, k) (OMPR, OMP, GOMP)"""
        self.call("csmp_solver_acquire", i64(int(k)))

    def solver_remove(self, atom):
        self.call("csmp_solver_remove", i64(int(atom)))

    def solver_state(self, cap):
        idx = np.zeros(cap, np.int64)
        val = np.zeros(cap, np.float64)
        order = np.zeros(cap, np.int64)
        nnz = i64(0)
        res = C.c_double(0)
        stop = C.c_int(0)
        self.call("csmp_solver_state", ptr(idx), ptr(val), C.byref(nnz), C.byref(res), ptr(order), C.byref(stop))
        n = nnz.value
        return idx[:n].copy(), val[:n].copy(), res.value, order[:n].copy(), stop.value

    # ---- one signal, columns sharded (csmp_shard_*): torch CUDA byte tensors for the records
    def set_option(self, key, value):
        """csmp_set_option: key is an OPT_* constant or its lower-case name ("batch_cert", "batch_gram", ...)."""
        self.call("csmp_set_option", int(OPTIONS.get(key, key)), int(value))

    def get_option(self, key):
        v = i64(0)
        self.call("csmp_get_option", int(OPTIONS.get(key, key)), C.byref(v))
        return int(v.value)

    def set_stream(self, hip_stream):
        """Borrow a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); 0 / None: the library's own."""
        self.call("csmp_set_stream", vp(hip_stream or 0))

    def shard_config(self, col_offset):
        self.call("csmp_shard_config", i64(int(col_offset)))

    def shard_record_bytes(self):
        return int(lib().csmp_shard_record_bytes(self._h))

    def shard_sweep(self, eps, check_eps, rec):
        assert rec.is_cuda and rec.is_contiguous() and rec.numel() * rec.element_size() >= self.shard_record_bytes()
        self.call("csmp_shard_sweep", C.c_double(eps), int(bool(check_eps)), vp(rec.data_ptr()))

    def shard_append(self, recs, nrec):
        assert recs.is_cuda and recs.is_contiguous() and recs.numel() * recs.element_size() >= nrec * self.shard_record_bytes()
        self.call("csmp_shard_append", vp(recs.data_ptr()), int(nrec))

    # ---- primitives
    def sweep(self, r, topk=1, want_abs=True):
        r = np.ascontiguousarray(r, dtype=np.float64)
        if r.shape != (self.M,):
            raise CsmpError(EDIM, "length(r) != size(A, 1)")
        out = np.zeros(self.N, np.float64) if want_abs else None
        ti = np.zeros(max(int(topk), 1), np.int64)
        tv = np.zeros(max(int(topk), 1), np.float64)
        self.call("csmp_sweep", ptr(r), ptr(out), i64(int(topk)), ptr(ti), ptr(tv))
        return out, ti[:int(topk)], tv[:int(topk)]

    def lstsq(self, cols, b):
        b = self._b(b)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        coef = np.zeros(len(cols), np.float64)
        self.call("csmp_lstsq", ptr(cols), i64(len(cols)), ptr(b), dtype_code(b.dtype), ptr(coef))
        return coef

    # ---- measurement
    def profile_enable(self, on=True):
        """on=True: time every sweep launch with HIP events; on=n>1: every n-th; False: off."""
        self.call("csmp_profile_enable", int(on))

    def profile_read(self, reset=True):
        n = i64(0)
        ms = C.c_double(0)
        self.call("csmp_profile_read", C.byref(n), C.byref(ms), int(bool(reset)))
        return n.value, ms.value

    def profile_window(self):
        """the timed launches of this context and its twin as one window (csmp_internal.h); call before profile_read"""
        n, w, d, st = i64(0), C.c_double(0), C.c_double(0), C.c_int(0)
        self.call("csmp_profile_window", C.byref(n), C.byref(w), C.byref(d), C.byref(st))
        return {"launches": int(n.value), "window_ms": w.value, "mean_launch_ms": d.value, "streams": st.value}

    def profile_overhead(self, reps=64):
        """average reading (ms) of an empty HIP-event pair on this context's stream"""
        ms = C.c_double(0)
        self.call("csmp_profile_overhead", int(reps), C.byref(ms))
        return ms.value

    def bench_sweep(self, variant=0, reps=20):
        ms = C.c_double(0)
        self.call("csmp_bench_sweep", int(variant), int(reps), C.byref(ms))
        return ms.value

    def sweep_config(self):
        """what configure_sweep chose for the resident dictionary (csmp_internal.h)"""
        unit, ph, wg, twg, lds, dyn, cpu = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), i64(0), C.c_int(0), C.c_int(0)
        self.call("csmp_sweep_config", C.byref(unit), C.byref(ph), C.byref(wg), C.byref(twg), C.byref(lds), C.byref(dyn), C.byref(cpu))
        return {"unit_loads": unit.value, "phases": ph.value, "workgroups": wg.value, "tick_workgroups": twg.value,
                "lds_bytes": int(lds.value), "dynamic": dyn.value, "columns_per_unit": cpu.value}

    def tune(self, key, value):
        """measurement override (csmp_internal.h): key in TUNE"""
        self.call("csmp_tune", int(TUNE[key]), i64(int(value)))


# ---- signal sharding: the wire layout of the ONE exchange (host memory, no ctx)
def shard_range(nsig, rank, world):
    lo, hi = i64(0), i64(0)
    rc = lib().csmp_shard_range(i64(int(nsig)), int(rank), int(world), C.byref(lo), C.byref(hi))
    if rc != OK:
        raise CsmpError(rc, "csmp_shard_range: bad arguments")
    return lo.value, hi.value


def pack_results(idx, val, nnz):
    """idx, val: (nsig, k) C-contiguous int64 / float64; nnz: (nsig,) int64 -> (nsig, 2k+1) float64 rows."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float64)
    nnz = np.ascontiguousarray(nnz, dtype=np.int64)
    nsig, k = idx.shape
    out = np.empty((nsig, 2 * k + 1), np.float64)
    rc = lib().csmp_pack_results(ptr(idx), ptr(val), ptr(nnz), i64(k), i64(nsig), ptr(out))
    if rc != OK:
        raise CsmpError(rc, "csmp_pack_results: bad arguments")
    return out


def unpack_results(packed, k):
    packed = np.ascontiguousarray(packed, dtype=np.float64)
    nsig = packed.shape[0]
    idx = np.empty((nsig, k), np.int64)
    val = np.empty((nsig, k), np.float64)
    nnz = np.empty(nsig, np.int64)
    rc = lib().csmp_unpack_results(ptr(packed), i64(int(k)), i64(nsig), ptr(idx), ptr(val), ptr(nnz))
    if rc != OK:
        raise CsmpError(rc, "csmp_unpack_results: bad arguments")
    return idx, val, nnz
