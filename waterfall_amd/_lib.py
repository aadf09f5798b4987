"""ctypes binding of libwfdwfst.so (the C ABI declared in include/wf_dwfst.h).

There is no CPU fallback: if the library has not been built (``python -c "import
__graft_entry__ as g; g.build()"``) importing a compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwfdwfst.so")

WF_DOUBLE_ACC = 1
WF_TROPICAL = 2
WF_KEEP_AB = 4
WF_PRUNED = 8
WF_PREZERO = 16
WF_ACCUMULATE = 32
WF_TOT_F64 = 64


class WfError(RuntimeError):
    pass


class _GraphHost(C.Structure):
    _fields_ = [
        ("num_fsas", C.c_int32),
        ("arcs", C.c_void_p),
        ("fsa_arc_offsets", C.c_void_p),
        ("fsa_num_states", C.c_void_p),
    ]


_lib = None

_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_u32 = C.c_uint32
_sz = C.c_size_t

_SIGS = {
    "wf_version": (C.c_int, []),
    "wf_last_error": (C.c_char_p, []),
    "wf_graph_prepare_bytes": (C.c_int, [C.POINTER(_GraphHost), C.POINTER(_sz)]),
    "wf_graph_prepare": (C.c_int, [C.POINTER(_GraphHost), _p, _sz]),
    "wf_compose_linear": (C.c_int, [_p, _p, _i32, _i32, _p, _i32, _p, _p, C.POINTER(_i32), C.POINTER(_i32), _i32]),
    "wf_compose_fsa_batch": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, C.c_int64, _p, _p]),
    "wf_loss_stats": (C.c_int, [_p, _i32, C.c_float, _p, _p]),
    "wf_label_grad_sums": (C.c_int, [_p, _i64, _i64, _i32, _i32, _i32, _p, _p]),
    "wf_topo_index_words": (C.c_int, [_p, _p, _i32, _i32, C.POINTER(C.c_int64)]),
    "wf_topo_index_build": (C.c_int, [_p, _p, _i32, _i32, _p]),
    "wf_compose_linear_batch": (C.c_int, [_p, _p, _i32, _i32, _p, _i32, _p, _p, _i32, _p, _p, _i32, _p, _p, C.c_int64, _p, _p]),
    "wf_init": (C.c_int, []),
    "wf_shutdown": (C.c_int, []),
    "wf_debug_reload_env": (C.c_int, []),
    "wf_debug_fallback_count": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _u32, _i32, _p, _sz, _p, C.POINTER(_i32)]),
    "wf_numerator_compile": (C.c_int, [_p, _p, _i32, _i32, _p, _i32, _p, _p, _i32, _p, _p, _i32, _p, _p, C.c_int64, _p, _p,
                                       _p, _sz, C.POINTER(_sz), _i32]),
    "wf_topo_device_build": (C.c_int, [_p, _p, _i32, _i32, _p, _i64, _i32, _p, _p, _i32, _p, _i64, C.POINTER(_i64)]),
    "wf_numerator_compile_device_bytes": (C.c_int, [_p, _p, _i32, C.POINTER(_sz), C.POINTER(_i64), C.POINTER(_sz)]),
    "wf_numerator_compile_device": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _sz, _p, _p, _i64, _p, _p, _sz, _p]),
    "wf_numerator_compile_job_bytes": (C.c_int, [_p, _p, _i32, C.POINTER(_sz), C.POINTER(_sz), _p]),
    "wf_numerator_compile_job": (C.c_int, [_p, _p, _p, _i32, _p, _sz, _p, _sz, _i32, _p, _p, _i64, C.POINTER(_i64), _p]),
    "wf_compile_job_release": (C.c_int, [_i32, _i64, _p, _i32]),
    "wf_compile_job_wait": (C.c_int, [_i32, _i64]),
    "wf_compile_job_join": (C.c_int, [_i32, _i64, _p]),
    "wf_fb_workspace_bytes": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _u32, _i32, C.POINTER(_sz)]),
    "wf_fb_pruned_is_fast": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _u32, C.POINTER(_i32)]),
    "wf_static_scratch_bytes": (C.c_int, [C.POINTER(_sz)]),
    "wf_debug_alloc_count": (C.c_int, [C.POINTER(C.c_int64)]),
    "wf_fb_logsum_static": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _i32, C.c_float, _u32,
                                      _p, _p, _p, _p]),
    "wf_fb_logsum": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _i32, C.c_float, _u32,
                               _p, _p, _p, _p, _sz, _p]),
    "wf_fb_backward": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _i32, _u32,
                                 _p, _p, _p, _p, _sz, _p]),
    "wf_loss_reduce": (C.c_int, [_p, _p, _i32, _p, _p]),
    "wf_grad_scale": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i64, _i64, _p, _i32, _p, _p, _sz, _p]),
    "wf_state_posteriors": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _u32, _p, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "wf_tot_scores": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _i32, _u32, _p, _p, _sz, _p]),
    "wf_timing_enable": (C.c_int, [C.c_int]),
    "wf_timing_read": (C.c_int, [C.POINTER(C.c_float)]),
    "wf_viterbi_workspace_bytes": (C.c_int, [_p, _p, _p, _i32, _i32, _u32, C.POINTER(_sz)]),
    "wf_viterbi": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, _u32, _p, _p, _p, _sz, _p]),
    "wf_search_workspace_bytes": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, C.POINTER(_sz)]),
    "wf_search": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, C.c_float, _i32, _i32, _u32, _p, _p, _p, _p,
                            _p, _sz, _p]),
    "wf_search_lattice_workspace_bytes": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, C.POINTER(_sz)]),
    "wf_search_lattice_words_workspace_bytes": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, C.POINTER(_sz)]),
    "wf_search_lattice_words": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, C.c_float, C.c_float, _i32,
                                          _i32, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "wf_search_lattice": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, C.c_float, C.c_float, _i32, _i32,
                                    _u32, _p, _p, _p, _p, _p, _p, _sz, _p]),
}


def lib():
    """Load the shared library once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WfError(
                f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "waterfall_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise WfError(f"libwfdwfst error {rc}: {lib().wf_last_error().decode()}")


ENV_EPOCH = 0


def reload_env():
    """Test aid: make the library re-read its WF_* environment hooks (read once per process otherwise) and invalidate
    the Python-side caches of plan-dependent sizes (Fsa.workspace_bytes)."""
    global ENV_EPOCH
    ENV_EPOCH += 1
    return lib().wf_debug_reload_env()


def exported_symbols():
    return list(_SIGS.keys())
