"""ctypes binding of libwcx_hip.so (include/wcx.h).

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible when
a context is created, the product path raises.  The library is built in-tree by
`__graft_entry__.build()` / `make -C wisecondorx_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwcx_hip.so")

c_i64 = C.c_int64
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/wcx.h declares.
SIGNATURES = {
    "wcx_version": (C.c_int, []),
    "wcx_format_bins_bed": (c_i64, [C.c_char_p, c_i64, c_i64, vp, vp, vp, c_i64]),
    "wcx_format_floats": (c_i64, [vp, c_i64, C.c_char, vp, c_i64]),
    "wcx_layout_counts": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]),
    "wcx_debug_flags": (C.c_int, [vp, C.c_int]),
    "wcx_sweep_event": (C.c_int, [vp, C.POINTER(vp)]),
    "wcx_wait_event": (C.c_int, [vp, vp]),
    "wcx_timer_tag": (C.c_int, [vp, C.c_char_p]),
    "wcx_predict_prep_dev": (C.c_int, [vp, vp, C.c_int, c_i64, vp, c_i64, vp, vp, C.c_int, vp]),
    "wcx_last_error": (C.c_char_p, []),
    "wcx_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "wcx_ctx_destroy": (C.c_int, [vp]),
    "wcx_sync": (C.c_int, [vp]),
    "wcx_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "wcx_free": (C.c_int, [vp, vp]),
    "wcx_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "wcx_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "wcx_last_kernel_ms": (C.c_double, [vp, C.c_char_p]),
    "wcx_last_topk_stats": (C.c_int, [vp, c_i64p]),
    "wcx_transpose_dev": (C.c_int, [vp, vp, c_i64, c_i64, vp]),
    "wcx_gather_transpose_dev": (C.c_int, [vp, vp, C.c_int, c_i64, c_i64, C.c_int, vp]),
    "wcx_compact_rows_dev": (C.c_int, [vp, vp, C.c_int, c_i64, c_i64, c_i64, vp]),
    "wcx_pca_begin": (C.c_int, [vp, vp, c_i64, C.c_int, vp, vp]),
    "wcx_pca_finish": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp]),
    "wcx_pca_end": (C.c_int, [vp]),
    "wcx_prep_mask_dev": (C.c_int, [vp, vp, c_i64, vp, C.c_int, vp, vp]),
    "wcx_pca_begin_counts_dev": (C.c_int, [vp, vp, c_i64, vp, C.c_int, c_i64, vp, c_i64, vp, vp]),
    "wcx_pca_corrected_dev": (C.c_int, [vp, C.POINTER(vp)]),
    "wcx_newref_topk": (C.c_int, [vp, vp, c_i64, C.c_int, c_i64p, C.c_int, c_i64, c_i64,
                                  C.c_int, C.c_int, vp, vp]),
    "wcx_newref_topk_dev": (C.c_int, [vp, vp, c_i64, C.c_int, c_i64p, C.c_int, c_i64, c_i64,
                                      C.c_int, C.c_int, vp, vp]),
    "wcx_null_ratios": (C.c_int, [vp, vp, c_i64, C.c_int, vp, c_i64, c_i64, C.c_int, c_i32p,
                                  C.c_int, vp]),
    "wcx_null_rank_prepare_dev": (C.c_int, [vp, vp, c_i64, C.c_int, c_i32p, C.c_int]),
    "wcx_null_ratios_dev": (C.c_int, [vp, vp, c_i64, C.c_int, vp, c_i64, c_i64, C.c_int,
                                      c_i32p, C.c_int, vp]),
    "wcx_ref_upload": (C.c_int, [vp, vp, vp, c_i64, C.c_int, c_i64p, C.c_int, C.POINTER(vp)]),
    "wcx_newref_sym_sweep_dev": (C.c_int, [vp, vp, c_i64, C.c_int, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          c_i64p, c_i64p]),
    "wcx_newref_sym_records_dev": (C.c_int, [vp, vp]),
    "wcx_newref_sym_finish_dev": (C.c_int, [vp, vp, c_i64, vp, vp]),
    "wcx_ref_wrap_dev": (C.c_int, [vp, vp, vp, c_i64, C.c_int, c_i64p, C.c_int, C.POINTER(vp)]),
    "wcx_ref_free": (C.c_int, [vp, vp]),
    "wcx_cutoff": (C.c_int, [vp, vp, C.c_int, c_f64p]),
    "wcx_weights": (C.c_int, [vp, vp, vp]),
    "wcx_predict_normalize": (C.c_int, [vp, vp, vp, C.c_int, C.c_double, c_i64, C.c_int, vp, vp,
                                        vp, vp, vp]),
    "wcx_predict_normalize_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_double, c_i64, C.c_int, vp,
                                            vp, vp, vp, vp]),
    "wcx_ref_wrap_rows_dev": (C.c_int, [vp, vp, vp, c_i64, C.c_int, c_i64p, C.c_int, c_i64, c_i64,
                                        C.POINTER(vp)]),
    "wcx_cutoff_moments_dev": (C.c_int, [vp, vp, C.c_double, C.c_double, C.c_int, c_f64p]),
    "wcx_predict_pass_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_double, c_i64, C.c_int, C.c_int, vp,
                                       vp, vp, vp]),
    "wcx_nanmedian2_dev": (C.c_int, [vp, vp, vp, c_i64, vp, vp]),
    "wcx_cbs": (C.c_int, [vp, vp, vp, c_i64p, C.c_int, C.c_double, c_i64, C.c_uint64, vp,
                          C.c_int, C.POINTER(C.c_int)]),
    "wcx_cbs_stats": (C.c_int, [vp, c_i64p]),
    "wcx_cbs_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, c_i64, c_i64p, C.c_int, C.c_double, c_i64, C.c_uint64,
                                    vp, C.c_int, vp]),
    "wcx_segment_z_dev": (C.c_int, [vp, vp, vp, c_i64p, C.c_int, vp, C.c_int, vp, vp]),
    "wcx_segment_z_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, c_i64p, C.c_int, vp, vp, vp, vp]),
    "wcx_post_process_merge_dev": (C.c_int, [vp, vp, vp, vp, vp, c_i64, vp, vp, vp, vp, c_i64, C.c_int, vp, vp,
                                             C.c_double, vp, c_i64, vp, vp, vp, vp]),
    "wcx_null_ratios_dummy_dev": (C.c_int, [vp, vp, c_i64, C.c_int, c_i64, c_i64, c_i32p, C.c_int, vp]),
    "wcx_cbs_trace": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "wcx_cbs_getbdry": (C.c_int, [C.c_double, C.c_int, C.c_int, vp]),
    "wcx_weights_dev": (C.c_int, [vp, vp, vp]),
    "wcx_post_process_dev": (C.c_int, [vp, vp, vp, vp, vp, c_i64, vp, vp, C.c_double, vp, c_i64, vp, vp, vp]),
    "wcx_cbs_batch": (C.c_int, [vp, vp, vp, C.c_int, c_i64, c_i64p, C.c_int, C.c_double, c_i64, C.c_uint64,
                                vp, C.c_int, vp]),
    "wcx_set_null_matrix": (C.c_int, [vp, vp, c_i64, C.c_int]),
    "wcx_set_null_matrix_dev": (C.c_int, [vp, vp, c_i64, C.c_int, vp, c_i64]),
    "wcx_segment_z": (C.c_int, [vp, vp, vp, vp, C.c_int, c_i64p, C.c_int, vp, C.c_int, vp, vp]),
}


WCX_ERR_UNSUPPORTED = 4        # include/wcx.h


class WcxError(RuntimeError):
    pass


_lib = None


def load():
    """Load libwcx_hip.so and bind every declared symbol.  Raises if absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WcxError(
            "{} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C wisecondorx_amd/csrc`. There is no CPU fallback.".format(LIB_PATH))
    # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Whichever
    # copy is loaded first serves the whole process; if ours came first, torch.cuda would later
    # fail with "No HIP GPUs are available".  So when torch is installed, let it load its runtime
    # first (the C-ABI itself does not depend on torch).
    if os.environ.get("WCX_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise WcxError("libwcx_hip error {}: {}".format(rc, load().wcx_last_error().decode()))


def ptr(a):
    """Raw pointer of a NumPy array (must stay alive for the call) or an int address."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(vp)
    return vp(int(a))


def i64_array(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.int64))
    return a, a.ctypes.data_as(c_i64p)


def i32_array(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.int32))
    return a, a.ctypes.data_as(c_i32p)


class Context:
    """One per (process, GPU).  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream, so launches order with the caller's work (0 = the
    default stream, which is what PyTorch uses unless told otherwise); None = a private stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = vp()
        if stream is None:
            arg = None                      # the library creates its own stream
        elif int(stream) == 0:
            arg = vp(1)                     # WCX_STREAM_DEFAULT: the null stream's handle is 0
        else:
            arg = vp(int(stream))
        check(self.lib.wcx_ctx_create(int(device), arg, C.byref(h)))
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.release_buffers()
            self.lib.wcx_ctx_destroy(self.h)
            self.h = None

    def buffers(self, sizes):
        """Device buffers kept between calls (grow-only; newref's three passes hand their result tables
        back through the same ones: a fresh hipMalloc of 0.8 GB after a hipFree costs ~60 ms).
        Returns one raw device pointer (int) per requested size."""
        keep = self.__dict__.setdefault("_buffers", [])
        while len(keep) < len(sizes):
            keep.append([vp(), 0])
        for slot, n in zip(keep, sizes):
            if slot[1] < n:
                if slot[0].value:
                    check(self.lib.wcx_free(self.h, slot[0]))
                    slot[0], slot[1] = vp(), 0
                check(self.lib.wcx_malloc(self.h, max(int(n), 8), C.byref(slot[0])))
                slot[1] = int(n)
        return [slot[0].value for slot in keep[:len(sizes)]]

    def release_buffers(self):
        for slot in self.__dict__.pop("_buffers", []):
            if slot[0].value and getattr(self, "h", None):
                self.lib.wcx_free(self.h, slot[0])

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.wcx_sync(self.h))

    def timer_tag(self, tag):
        check(self.lib.wcx_timer_tag(self.h, (tag or "").encode()))

    def kernel_ms(self, name):
        return float(self.lib.wcx_last_kernel_ms(self.h, name.encode()))

    def cbs_stats(self):
        out = (C.c_int64 * 4)()
        check(self.lib.wcx_cbs_stats(self.h, out))
        return {"bound_shortcuts": out[0], "arc_pairs_listed": out[1], "arc_pairs_total": out[2]}

    def cbs_trace(self):
        """Per-test records of the last CBS call (needs wcx_debug_flags(h, 128) before it); rows of
        20 doubles, see include/wcx.h."""
        n = C.c_int()
        check(self.lib.wcx_cbs_trace(self.h, None, 0, C.byref(n)))
        out = np.empty((n.value, 20))
        if n.value:
            check(self.lib.wcx_cbs_trace(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def topk_stats(self):
        out = (C.c_int64 * 24)()
        check(self.lib.wcx_last_topk_stats(self.h, out))
        return {"rows": out[0], "pairs": out[1], "compactions": out[2], "fallback_rows": out[3],
                "appends": out[4], "refined": out[5], "sym_gates": out[6], "sym_row_appends": out[7], "phase_cycles": [out[8 + i] for i in range(6)],
                "hub_trial_sum": out[16], "hub_rows_without_estimate": out[17], "hub_second_attempt_rows": out[18]}


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
