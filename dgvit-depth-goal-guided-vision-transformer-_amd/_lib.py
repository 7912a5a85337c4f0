"""ctypes binding of libdgvit_hip.so (C ABI: include/dgvit_hip.h).

There is no CPU or PyTorch fallback: if the shared library is missing or a call fails, a
``DgvitError`` is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdgvit_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "libdgvit_hip_diag.so")   # same sources built with -DDGVIT_DIAG: knobs and experiments (tools/, A/B tests)


class DgvitError(RuntimeError):
    pass


class dgvit_config(Structure):
    _fields_ = [("image_h", c_int), ("image_w", c_int), ("patch_h", c_int), ("patch_w", c_int), ("dim", c_int),
                ("depth", c_int), ("heads", c_int), ("dim_head", c_int), ("mlp_dim", c_int), ("pool_mean", c_int), ("flags", c_int)]


FLAG_DENSE_LAST_BLOCK = 1    # include/dgvit_hip.h: DGVIT_FLAG_*
FLAG_WGRAD_OVERLAP = 2


class dgvit_mlp_desc(Structure):
    _fields_ = [("batch", c_int), ("nseg", c_int), ("kx", c_int * 3), ("ldx", c_int * 3), ("n1", c_int), ("n2", c_int), ("n3", c_int),
                ("towers", c_int), ("heads3", c_int)]


class dgvit_grad_events(Structure):
    """include/dgvit_hip.h: events recorded by dgvit_got_backward[_bf16]_ev where a group of gradients is final"""
    _fields_ = [("n_layers", c_int), ("layer", POINTER(c_void_p)), ("head", c_void_p)]


NUM_GLOBAL_PARAMS = 4
PARAMS_PER_LAYER = 11
ABI_VERSION = 7

_P, _I, _LL, _F, _ULL = c_void_p, c_int, c_longlong, c_float, c_ulonglong
_CFG = POINTER(dgvit_config)
_TABLE = POINTER(c_void_p)

# name -> (restype, argtypes); every symbol include/dgvit_hip.h declares
SIGNATURES = {
    "dgvit_abi_version": (_I, []),
    "dgvit_config_size": (_I, []),
    "dgvit_last_error": (c_char_p, []),
    "dgvit_device_count": (_I, []),
    "dgvit_got_workspace_floats": (_LL, [_CFG, _I, _I]),
    "dgvit_got_backward_scratch_floats": (_LL, [_CFG, _I]),
    "dgvit_got_forward": (_I, [_CFG, _TABLE, _P, _P, _P, _P, _LL, _I, _I, _F, _ULL, _P, _P]),
    "dgvit_got_backward": (_I, [_CFG, _TABLE, _TABLE, _P, _P, _P, _LL, _P, _LL, _I, _F, _ULL, _P, _P]),
    "dgvit_got_backward_ev": (_I, [_CFG, _TABLE, _TABLE, _P, _P, _P, _LL, _P, _LL, _I, _F, _ULL, _P, _P, POINTER(dgvit_grad_events)]),
    "dgvit_event_create": (_I, [POINTER(c_void_p)]),
    "dgvit_event_destroy": (_I, [_P]),
    "dgvit_stream_wait_event": (_I, [_P, _P]),
    "dgvit_linear_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dgvit_linear_backward_scratch_floats": (_LL, [_I, _I, _I]),
    "dgvit_linear_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _P]),
    "dgvit_mlp_head_forward": (_I, [POINTER(dgvit_mlp_desc), _TABLE, _TABLE, _P, _P, _P, _P]),
    "dgvit_mlp_head_backward_scratch_floats": (_LL, [POINTER(dgvit_mlp_desc)]),
    "dgvit_mlp_head_backward": (_I, [POINTER(dgvit_mlp_desc), _TABLE, _TABLE, _P, _P, _TABLE, _TABLE, _TABLE, _P, _LL, _P]),
    "dgvit_tanh_gaussian_forward": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _I, _I, _P]),
    "dgvit_tanh_gaussian_backward": (_I, [_P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dgvit_gemm_scratch_floats": (_LL, [_I, _I, _I, _I]),
    "dgvit_gemm": (_I, [_I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P, _LL, _P]),
    "dgvit_layernorm_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dgvit_layernorm_backward_scratch_floats": (_LL, [_I, _I]),
    "dgvit_layernorm_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _P]),
    "dgvit_rmsnorm_forward": (_I, [_P, _LL, _P, _P, _I, _I, _P]),
    "dgvit_rmsnorm_backward_scratch_floats": (_LL, [_I, _I]),
    "dgvit_rmsnorm_backward": (_I, [_P, _P, _LL, _P, _P, _LL, _P, _P, _LL, _I, _I, _P]),
    "dgvit_attention_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dgvit_attention_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dgvit_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dgvit_dropout": (_I, [_P, _LL, _ULL, _F, _P]),
    "dgvit_cnn_workspace_floats": (_LL, [_I, _I, _I]),
    "dgvit_cnn_forward_scratch_floats": (_LL, [_I, _I, _I]),
    "dgvit_cnn_backward_scratch_floats": (_LL, [_I, _I, _I]),
    "dgvit_cnn_forward": (_I, [_P, _TABLE, _P, _P, _LL, _P, _LL, _I, _I, _I, _P]),
    "dgvit_cnn_backward": (_I, [_P, _TABLE, _TABLE, _P, _P, _LL, _P, _LL, _I, _I, _I, _P]),
    "dgvit_gather_rows": (_I, [_P, _P, _P, _LL, _LL, _LL, _P]),
    "dgvit_depth_preprocess_scratch_floats": (_LL, [_I, _I, _I]),
    "dgvit_depth_to_state": (_I, [_P, _P, _F, _ULL, _P, _P, _LL, _I, _I, _I, _I, _I, _P]),
    "dgvit_depth_normalize_u8": (_I, [_P, _P, _P, _LL, _I, _I, _I, _P]),
    "dgvit_noise_clip": (_I, [_P, _P, _P, _LL, _F, _ULL, _P]),
    "dgvit_gaussian_blur": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dgvit_resize_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "dgvit_adam_step": (_I, [_P, _P, _P, _P, _LL, _F, _F, _F, _F, _F, _LL, _P, _P]),
    "dgvit_soft_update": (_I, [_P, _P, _LL, _F, _P]),
    "dgvit_got_bf16_weight_elems": (_LL, [_CFG]),
    "dgvit_got_pack_weights_bf16": (_I, [_CFG, _TABLE, _P, _LL, _I, _P]),
    "dgvit_got_bf16_workspace_bytes": (_LL, [_CFG, _I, _I]),
    "dgvit_got_forward_bf16": (_I, [_CFG, _TABLE, _P, _P, _P, _P, _P, _LL, _I, _I, _F, _ULL, _P, _P]),
    "dgvit_got_bf16_backward_scratch_bytes": (_LL, [_CFG, _I]),
    "dgvit_got_backward_bf16": (_I, [_CFG, _TABLE, _P, _TABLE, _P, _P, _P, _P, _LL, _P, _LL, _I, _F, _ULL, _P, _P]),
    "dgvit_got_backward_bf16_ev": (_I, [_CFG, _TABLE, _P, _TABLE, _P, _P, _P, _P, _LL, _P, _LL, _I, _F, _ULL, _P, _P, POINTER(dgvit_grad_events)]),
    "dgvit_wgrad_bf16_scratch_floats": (_LL, [_I, _I, _I]),
    "dgvit_wgrad_bf16": (_I, [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _P]),
    "dgvit_cast_f32_bf16": (_I, [_P, _P, _LL, _P]),
    "dgvit_gemm_bf16": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    "dgvit_layernorm_forward_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dgvit_attention_forward_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dgvit_attention_backward_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dgvit_profile_start": (_I, [_I]),
    "dgvit_profile_stop": (_I, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_longlong)]),
    "dgvit_profile_sampling": (_I, [_I]),
    "dgvit_profile_totals": (_I, [POINTER(ctypes.c_double), POINTER(c_longlong)]),
}
# the additional entry points of libdgvit_hip_diag.so (include/dgvit_hip_diag.h)
DIAG_SIGNATURES = {
    "dgvit_set_gemm_tile": (None, [_I]),
    "dgvit_set_grouped_reduce": (None, [_I]),
    "dgvit_set_gemm_split": (None, [_I]),
    "dgvit_set_gemm_lds_pad": (None, [_I]),
    "dgvit_set_ln_fusion": (None, [_I]),
    "dgvit_set_conv_gather": (None, [_I]),
    "dgvit_set_gemm_diagnostics": (None, [_I]),
    "dgvit_set_gemm_persistent": (None, [_I, _I]),
    "dgvit_gemm_persistent_launches": (ctypes.c_longlong, []),
    "dgvit_set_gemm_stamps": (None, [_P, _I]),
    "dgvit_set_small_batch_path": (None, [_I, _I]),
    "dgvit_set_block_path": (None, [_I, _I]),
    "dgvit_set_block_stamps": (None, [_P]),
    "dgvit_set_block_stamp_layer": (None, [_I]),
    "dgvit_set_gelu_grad_store": (None, [_I]),
    "dgvit_set_block_fuse": (None, [_I]),
    "dgvit_set_gemm_bf16_tile": (None, [_I]),
    "dgvit_set_gemm_bf16_group_m": (None, [_I]),
    "dgvit_set_gemm_bf16_l2_budget_kb": (None, [_I]),
    "dgvit_set_attention_bwd_single_pass": (None, [_I]),
    "dgvit_set_attention_single_query": (None, [_I]),
    "dgvit_set_gemm_wgrad_slice_major": (None, [_I]),
    "dgvit_attention_forward_queries": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dgvit_attention_backward_queries": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dgvit_set_gemm_bf16_mfma16": (None, [_I]),
    "dgvit_set_gemm_bf16_stamps": (None, [_P]),
}
PROFILE_KINDS = 4

_lib = None        # the library the package's calls go through (the product library unless diagnostic() is active)
_product = None
_diag = None


def _open(path, signatures):
    if not os.path.exists(path):
        raise DgvitError(f"{path} not found: build it with `python {os.path.join(_HERE, 'build.py')}` "
                         "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DgvitError(f"{path} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    if lib.dgvit_abi_version() != ABI_VERSION:
        raise DgvitError(f"ABI mismatch: library {lib.dgvit_abi_version()} != binding {ABI_VERSION}")
    if lib.dgvit_config_size() != ctypes.sizeof(dgvit_config):
        raise DgvitError(f"dgvit_config is {ctypes.sizeof(dgvit_config)} bytes in the binding, {lib.dgvit_config_size()} in the library")
    return lib


def load():
    """The library every call of the package goes through: libdgvit_hip.so (loaded and typed once; DgvitError if it is missing or
    stale), or the diagnostic build while a ``diagnostic()`` block is active."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if _product is None:
        _product = _open(LIB_PATH, SIGNATURES)
    _lib = _product
    return _lib


class diagnostic:
    """``with diagnostic() as lib:`` routes the package through libdgvit_hip_diag.so (the -DDGVIT_DIAG build with the A/B knobs
    and experiments of include/dgvit_hip_diag.h) and yields it; the product library is back on exit.  Not thread-safe; for
    tools/ and the A/B equality tests, never for the product path."""

    def __enter__(self):
        global _lib, _diag
        load()
        if _diag is None:
            _diag = _open(DIAG_LIB_PATH, {**SIGNATURES, **DIAG_SIGNATURES})
        self._prev = _lib
        _lib = _diag
        return _diag

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def check(rc, what):
    if rc != 0:
        msg = load().dgvit_last_error()
        raise DgvitError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
