"""ctypes binding of the C ABI declared in include/fatezero_hip.h.

The product path loads exactly one library: the in-tree ``libfatezero_hip.so`` built by hipcc for gfx950
(``python -m fatezero_amd.build``).  If it is missing or does not load, every op raises -- there is no
CPU fallback.  ``use_test_backend`` exists only so the GPU-less test-suite can point the very same host code at
``libfatezero_emu.so`` (the kernel sources compiled against the CPU emulation of csrc/fz_rt.h).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(HERE, "libfatezero_hip.so")

FZ_ATTN_FLASH, FZ_ATTN_CAPTURE, FZ_ATTN_INJECT = 0, 1, 2
FZ_MAX_KV_SLOTS = 4
FZ_CROSS_MAX_KEYS = 96
FZ_CROSS_P_STRIDE = 80


class FzAttnSelfDesc(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("frame0", C.c_int32), ("clip_len", C.c_int32), ("heads", C.c_int32),
        ("head_dim", C.c_int32), ("lq", C.c_int32), ("lkf", C.c_int32), ("n_kv", C.c_int32),
        ("kv_abs", C.c_int32 * FZ_MAX_KV_SLOTS), ("kv_val", C.c_int32 * FZ_MAX_KV_SLOTS),
        ("scale", C.c_float), ("mode", C.c_int32),
        ("q_frame_stride", C.c_int64), ("q_row_stride", C.c_int64),
        ("k_frame_stride", C.c_int64), ("k_row_stride", C.c_int64),
        ("vt_frame_stride", C.c_int64), ("vt_chan_stride", C.c_int64),
        ("o_frame_stride", C.c_int64), ("o_row_stride", C.c_int64),
        ("p_frame_stride", C.c_int64), ("p_head_stride", C.c_int64), ("p_row_stride", C.c_int64),
        ("p_frame_off", C.c_int32), ("mask_frame_off", C.c_int32),
        ("k_head_stride", C.c_int64),
        ("q_log2_scaled", C.c_int32), ("kv_clip_len", C.c_int32), ("kv_frame_off", C.c_int32), ("reserved0", C.c_int32),
    ]


class FzAttnCrossDesc(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("frame0", C.c_int32), ("clip_len", C.c_int32), ("heads", C.c_int32),
        ("head_dim", C.c_int32), ("lq", C.c_int32), ("lk", C.c_int32),
        ("scale", C.c_float), ("mode", C.c_int32),
        ("q_frame_stride", C.c_int64), ("q_row_stride", C.c_int64),
        ("k_batch_stride", C.c_int64), ("k_row_stride", C.c_int64),
        ("vt_batch_stride", C.c_int64), ("vt_chan_stride", C.c_int64),
        ("o_frame_stride", C.c_int64), ("o_row_stride", C.c_int64),
        ("p_frame_stride", C.c_int64), ("p_head_stride", C.c_int64), ("p_row_stride", C.c_int64),
        ("p_frame_off", C.c_int32), ("store_cur", C.c_int32),
    ]


class FzGemmDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_int64), ("rows_store", C.c_int64), ("in_features", C.c_int32), ("out_features", C.c_int32),
        ("ldx", C.c_int64), ("ldw", C.c_int64), ("ldy", C.c_int64), ("ldres", C.c_int64),
        ("batch", C.c_int32), ("epilogue", C.c_int32),
        ("x_batch_stride", C.c_int64), ("y_batch_stride", C.c_int64), ("res_batch_stride", C.c_int64),
        ("transpose_out", C.c_int32), ("tile_cfg", C.c_int32), ("split_k", C.c_int32), ("reserved0", C.c_int32),
        ("workspace_floats", C.c_int64), ("w_batch_stride", C.c_int64),
    ]


class FzGemmLn(C.Structure):
    _fields_ = [("stats_in", C.c_void_p), ("c1", C.c_void_p), ("c0", C.c_void_p), ("eps", C.c_float), ("reserved0", C.c_int32),
                ("stats_out", C.c_void_p)]


class FzXattnChain(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("res", C.c_void_p), ("packed", C.c_void_p), ("kv_packed", C.c_void_p), ("bias_out", C.c_void_p),
        ("y", C.c_void_p), ("y_ln", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("y1", C.c_void_p),
        ("rows", C.c_int64), ("rows_per_frame", C.c_int64),
        ("frames_per_batch", C.c_int32), ("channels", C.c_int32), ("heads", C.c_int32), ("lk", C.c_int32),
        ("scale", C.c_float), ("ln_eps", C.c_float), ("ln1_eps", C.c_float), ("front", C.c_int32),
    ]


FZ_GEMM_PLAIN, FZ_GEMM_GEGLU = 0, 1
FZ_GEMM_NO_STATS = 1

_P = C.c_void_p
_SIGS = {
    "fz_gemm": (C.c_int, [C.POINTER(FzGemmDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "fz_ff_chain_ok": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "fz_ff_chain_preferred": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "fz_ff_chain_pack_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "fz_ff_chain_pack": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "fz_ff_chain": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "fz_xattn_chain_ok": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "fz_xattn_chain_preferred": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "fz_xattn_chain_pack_bytes": (C.c_int64, [C.c_int]),
    "fz_xattn_chain_kv_pack_bytes": (C.c_int64, [C.c_int]),
    "fz_xattn_chain_pack": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "fz_xattn_chain_kv_pack": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _P, _P]),
    "fz_xattn_chain": (C.c_int, [C.POINTER(FzXattnChain), _P]),
    "fz_gemm_lnout": (C.c_int, [C.POINTER(FzGemmDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_int64, _P, _P]),
    "fz_gemm_ln": (C.c_int, [C.POINTER(FzGemmDesc), C.POINTER(FzGemmLn), _P, _P, _P, _P, _P, _P, _P, _P]),
    "fz_gemm_qkvt": (C.c_int, [C.POINTER(FzGemmDesc), _P, _P, _P, _P, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P]),
    "fz_gn_epilogue_chunks": (C.c_int, [C.c_int64]),
    "fz_gemm_gn": (C.c_int, [C.POINTER(FzGemmDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int64, _P]),
    "fz_temporal_conv3_gn": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64,
                                       _P, C.c_int, _P]),
    "fz_groupnorm_from_partials": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P,
                                             C.c_int, _P, _P]),
    "fz_gemm_workspace_floats": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "fz_attn_self": (C.c_int, [C.POINTER(FzAttnSelfDesc), _P, _P, _P, _P, _P, _P, _P]),
    "fz_attn_cross": (C.c_int, [C.POINTER(FzAttnCrossDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fz_attn_temporal": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                   C.c_int64, C.c_float, _P]),
    "fz_attn_temporal_ex": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      C.c_int64, C.c_int64, C.c_float, _P]),
    "fz_blend_mask": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, _P,
                                C.c_float, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "fz_groupnorm_chunks": (C.c_int, [C.c_int, C.c_int]),
    "fz_groupnorm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                               _P, _P]),
    "fz_groupnorm_cat": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   _P, _P]),
    "fz_groupnorm_stats": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "fz_groupnorm_apply": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P,
                                     C.c_int, C.c_int, _P, _P]),
    "fz_conv3x3_up2_ok": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fz_conv3x3_up2_preferred": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fz_conv3x3_up2_pack_halves": (C.c_int64, [C.c_int, C.c_int]),
    "fz_conv3x3_up2_pack": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "fz_conv3x3_up2": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "fz_conv3x3": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "fz_temporal_conv3": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64,
                                    _P]),
    "fz_lora_pair_ok": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fz_lora_pair_preferred": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fz_lora_pair_gn_chunks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fz_lora_pair_gn": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "fz_lora_pair": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "fz_layernorm": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "fz_geglu": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "fz_softmax_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_float, _P]),
    "fz_transpose_pad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, _P]),
    "fz_latent_update": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P, C.c_int, C.c_int, _P]),
    "fz_accumulate": (C.c_int, [_P, _P, C.c_int64, _P]),
    "fz_peer_put": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int, C.c_uint32, _P, _P]),
    "fz_peer_wait": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_int64, _P]),
    "fz_plan_begin": (C.c_int, [C.POINTER(_P)]),
    "fz_plan_pause": (C.c_int, [_P, C.c_int]),
    "fz_plan_end": (C.c_int, [_P]),
    "fz_plan_launches": (C.c_int64, [_P]),
    "fz_plan_relocate": (C.c_int64, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P]),
    "fz_plan_replay": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "fz_plan_graph_launch": (C.c_int, [_P, _P]),
    "fz_plan_destroy": (None, [_P]),
    "fz_version": (C.c_char_p, []),
}

_lib = None
_lib_path = None
_is_test_backend = False


class NativeLibraryError(RuntimeError):
    pass


def _open(path):
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} is missing: build it with `python -m fatezero_amd.build` (hipcc --offload-arch=gfx950). "
            "The FateZero hot path has no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 not loadable
        raise NativeLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    return lib


def lib():
    """The loaded native library (HIP build unless a test explicitly switched the backend)."""
    global _lib, _lib_path
    if _lib is None:
        _lib = _open(HIP_LIB)
        _lib_path = HIP_LIB
    return _lib


def use_test_backend(path):
    """TEST HOOK: run the host code against the CPU emulation build. Never called by the product."""
    global _lib, _lib_path, _is_test_backend
    _lib = _open(path)
    _lib_path = path
    _is_test_backend = True


def reset_backend():
    global _lib, _lib_path, _is_test_backend
    _lib, _lib_path, _is_test_backend = None, None, False


def is_test_backend():
    return _is_test_backend


def loaded_path():
    return _lib_path


def exported_symbols():
    return list(_SIGS.keys())


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")
