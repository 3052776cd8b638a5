"""ctypes binding of libsafe_hip.so (the C ABI declared in include/safe_hip.h).

There is no fallback: if the shared library is missing or cannot be loaded, importing
this module raises, and every compute call raises ``SafeHipError`` when the library
reports a failure (for example when no HIP device is present).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsafe_hip.so')

ABI_VERSION = 3
DTYPE_F32, DTYPE_F64, DTYPE_U8 = 0, 1, 2
SCORE_SUM, SCORE_ZSCORE = 0, 1
SIGN_HIGHEST, SIGN_LOWEST, SIGN_BOTH = 0, 1, 2
E_INVALID, E_HIP, E_NOMEM, E_UNSUPPORTED, E_VALUE = -1, -2, -3, -4, -5


class SafeHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__('libsafe_hip error %d: %s' % (code, message))
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        'safepy_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` or '
        '`make -C safepy_amd/csrc` (needs hipcc, --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)



def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (same SONAME as /opt/rocm's).  If libsafe_hip.so were loaded first it would bind to
    /opt/rocm's copy and a later `import torch` would bring a second runtime into the
    process: device pointers, streams and torch.cuda.synchronize() would then belong to
    different runtimes.  Loading torch's copy first (without importing torch) makes the
    dynamic loader resolve libsafe_hip.so's NEEDED libamdhip64.so.7 to it, and torch
    later finds the very same file.  Without torch installed, /opt/rocm's runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    # the same rule for RCCL, lazily: safe_comm_* (comm.cpp) dlopens RCCL on first use and must pick the copy a later
    # `import torch` would bring (two RCCL copies in one process corrupt the heap at exit)
    rccl = os.path.join(libdir, 'librccl.so')
    if os.path.exists(rccl):
        os.environ.setdefault('SAFE_HIP_RCCL_PATH', rccl)
    path = os.path.join(libdir, 'libamdhip64.so')
    if not os.path.exists(path):
        return None
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


HIP_RUNTIME_PRELOADED = _preload_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)

_vp = C.c_void_p
_i64 = C.c_int64
_pp = C.POINTER(C.c_void_p)
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/safe_hip.h declares
PROTOTYPES = {
    'safe_abi_version': (C.c_int, []),
    'safe_last_error': (C.c_char_p, []),
    'safe_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'safe_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    'safe_ctx_create': (C.c_int, [C.c_int, _pp]),
    'safe_ctx_destroy': (C.c_int, [_vp]),
    'safe_ctx_set_stream': (C.c_int, [_vp, _vp]),
    'safe_ctx_sync': (C.c_int, [_vp]),
    'safe_ctx_info': (C.c_int, [_vp, C.POINTER(C.c_int), _pi64, C.c_char_p, C.c_size_t]),
    'safe_dev_alloc': (C.c_int, [_vp, C.c_size_t, _pp]),
    'safe_dev_free': (C.c_int, [_vp, _vp]),
    'safe_dev_memset': (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    'safe_memcpy_h2d': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'safe_memcpy_d2h': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'safe_memcpy_d2h_resident': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'safe_timer_start': (C.c_int, [_vp]),
    'safe_timer_stop_ms': (C.c_int, [_vp, C.POINTER(C.c_double)]),
    'safe_nbr_euclidean': (C.c_int, [_vp, _vp, _i64, C.c_double, _pp]),
    'safe_nbr_shortpath': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, C.c_double, C.c_int, _pp]),
    'safe_nbr_from_dense_i64': (C.c_int, [_vp, _vp, _i64, _pp]),
    'safe_nbr_set_layout': (C.c_int, [_vp, _vp]),
    'safe_nbr_block_count': (C.c_int, [_vp, _pi64]),
    'safe_nbr_piece_count': (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    'safe_nbr_destroy': (C.c_int, [_vp]),
    'safe_nbr_info': (C.c_int, [_vp, _pi64, _pi64, _pi64]),
    'safe_nbr_to_dense_i64': (C.c_int, [_vp, _vp]),
    'safe_nbr_to_dense_i64_dev': (C.c_int, [_vp, _vp]),
    'safe_nbr_row_counts': (C.c_int, [_vp, _vp]),
    'safe_nbr_csr': (C.c_int, [_vp, _vp, _vp]),
    'safe_nbr_distances': (C.c_int, [_vp, _vp]),
    'safe_euclidean_dense_dev': (C.c_int, [_vp, _vp, _i64, C.c_double, _vp, _vp]),
    'safe_edge_lengths': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'safe_attr_create_host': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _i64, _pp]),
    'safe_attr_create_dev': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _i64, _pp]),
    'safe_attr_destroy': (C.c_int, [_vp]),
    'safe_attr_reindex': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _i64, _vp, _i64, C.c_double, C.c_int, _vp, _pp]),
    'safe_attr_value_counts': (C.c_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    'safe_attr_nan_to_zero': (C.c_int, [_vp]),
    'safe_attr_download': (C.c_int, [_vp, _vp]),
    'safe_attr_stats': (C.c_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    'safe_attr_row_flags': (C.c_int, [_vp, _vp]),
    'safe_attr_set_row_flags': (C.c_int, [_vp, _vp]),
    'safe_perms_create': (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, C.c_uint32, _pp]),
    'safe_perms_destroy': (C.c_int, [_vp]),
    'safe_perms_read': (C.c_int, [_vp, _i64, _i64, _vp]),
    'safe_rng_permutations_host': (C.c_int, [C.c_uint32, _vp, _i64, _i64, _vp]),
    'safe_score': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _i64, _vp]),
    'safe_permtest_counts': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _i64, _i64, _vp, _vp, _vp]),
    'safe_randomization': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_double, _vp, _i64, _i64,
                                     _vp, _vp, _vp, _vp, _vp, _vp]),
    'safe_hypergeom': (C.c_int, [_vp, _vp, _vp, C.c_double, _i64, _i64, _vp, _vp, _vp, _vp]),
    'safe_enriched_components': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    'safe_jaccard_condensed': (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
    'safe_fdr_adjust': (C.c_int, [_vp, _i64, _i64, _i64, C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    'safe_export_packed_counts': (C.c_int, [_vp, _vp, _i64, _pi64, _pi64, C.POINTER(C.c_int)]),
    'safe_nes_from_packed_counts': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _i64, _i64, C.c_int, _vp, _vp]),
    'safe_outputs_from_packed_counts': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _i64, _i64, C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    'safe_set_exchange_chunks': (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    'safe_packed_chunk_info': (C.c_int, [_vp, C.POINTER(C.c_int), _pi64, _pi64]),
    'safe_export_packed_chunk': (C.c_int, [_vp, C.c_int, _vp, _i64, _vp]),
    'safe_export_packed_chunk_narrow': (C.c_int, [_vp, C.c_int, _vp, _i64, _vp]),
    'safe_outputs_from_packed_slabs': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, _i64, _pi64, _pi64, _i64, _i64, C.c_int, C.c_double,
                                                 _vp, _vp, _vp, _vp, _vp, _vp]),
    'safe_randomization_plan': (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.POINTER(C.c_int)]),
    'safe_last_kernel_stats': (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), _pi64]),
    'safe_last_kernel_busy_ms': (C.c_int, [_vp, C.POINTER(C.c_double)]),
    'safe_last_mfma_slices': (C.c_int, [_vp, C.POINTER(C.c_int)]),
    'safe_last_mfma_filter': (C.c_int, [_vp, C.POINTER(C.c_int), _pi64]),
    'safe_alloc_count': (C.c_int, [C.POINTER(C.c_int64)]),
    'safe_build_info': (C.c_int, [C.c_char_p, C.c_size_t]),
    'safe_set_draw_cpus': (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    'safe_perms_create_from_table': (C.c_int, [_vp, _i64, _i64, _vp, _pp]),
    'safe_perms_slice': (C.c_int, [_vp, _i64, _i64, _pp]),
    'safe_outputs_from_counts': (C.c_int, [_vp, _i64, _i64, _i64, C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'safe_comm_unique_id': (C.c_int, [C.c_char_p, C.c_size_t]),
    'safe_comm_create': (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.c_size_t, _pp]),
    'safe_comm_destroy': (C.c_int, [_vp]),
    'safe_allgather_cols': (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    'safe_set_blocking_sync': (C.c_int, [C.c_int]),
    'safe_ctx_share_stream': (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, _i64]),
    'safe_ctx_unshare_stream': (C.c_int, [_vp]),
    'safe_perms_create_shared': (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, C.c_uint32, _pp]),
    'safe_perms_create_device': (C.c_int, [_vp, _i64, _vp, _i64, C.c_uint64, _pp]),
    'safe_perms_timing': (C.c_int, [_vp, C.POINTER(C.c_double)]),
    'safe_perms_twin_stats': (C.c_int, [_vp, C.POINTER(C.c_int), _pi64, _pi64]),
    'safe_ring_open': (C.c_int, [C.c_char_p, C.c_int, C.c_int, _i64, _pp]),
    'safe_ring_close': (C.c_int, [_vp]),
    'safe_ring_begin': (C.c_int, [_vp, _i64, _i64, _i64, C.c_uint64, _i64]),
    'safe_ring_publish': (C.c_int, [_vp, _i64, _vp, C.c_size_t]),
    'safe_ring_fetch': (C.c_int, [_vp, _i64, _vp, C.c_size_t]),
    'safe_ring_end': (C.c_int, [_vp]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)      # AttributeError here = the library does not export the ABI
    _fn.restype = _res
    _fn.argtypes = _args

if lib.safe_abi_version() != ABI_VERSION:
    raise ImportError('safepy_amd: libsafe_hip.so ABI %d != expected %d; rebuild it'
                      % (lib.safe_abi_version(), ABI_VERSION))


def check(code):
    if code != 0:
        raise SafeHipError(code, lib.safe_last_error().decode('utf-8', 'replace'))


def device_count():
    c = C.c_int(0)
    check(lib.safe_device_count(C.byref(c)))
    return c.value
