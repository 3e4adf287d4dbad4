"""ctypes binding of libdvla_hip.so (C ABI declared in include/dvla.h).

The product path has NO CPU / eager fallback: if the shared library is missing or a kernel is asked to run
on a non-CUDA tensor the call raises.  (`oracle/` holds the CPU restatement used only as a checker.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVLA_LIB: load another build of the library (same-box A/B measurements of kernel changes; not a product knob)
LIB_PATH = os.environ.get("DVLA_LIB") or os.path.join(_HERE, "libdvla_hip.so")

DT_BF16, DT_F32 = 0, 1
ABI_VERSION = 8          # DVLA_ABI_VERSION of include/dvla.h
ACT = {"none": 0, "gelu": 1, "gelu_erf": 1, "gelu_tanh": 2, "gelu_new": 2, "relu": 3, "silu": 4,
       "quick_gelu": 5, "tanh": 6, "sigmoid": 7}

DVLA_ERRORS = {-1: "invalid argument", -2: "kernel launch failed", -3: "unsupported shape/alignment"}


class DvlaError(RuntimeError):
    pass


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64), ("a_trans", C.c_int32),
        ("B", C.c_void_p), ("ldb", C.c_int64), ("b_trans", C.c_int32),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_dtype", C.c_int32),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("bias", C.c_void_p), ("bias_dtype", C.c_int32),
        ("act", C.c_int32),
        ("preact", C.c_void_p), ("ld_preact", C.c_int64),
        ("dact_aux", C.c_void_p), ("ld_dact", C.c_int64), ("dact", C.c_int32),
        ("dropout_p", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("residual", C.c_void_p), ("ld_res", C.c_int64), ("res_rows", C.c_int64),
        ("accumulate", C.c_int32),
        ("split_k", C.c_int32), ("workspace", C.c_void_p),
        ("ksum", C.c_void_p), ("ksum_dtype", C.c_int32), ("ksum_operand", C.c_int32), ("ksum_workspace", C.c_void_p),
        ("a_layernorm", C.c_int32), ("a_ln_eps", C.c_float),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("q_stride_b", C.c_int64), ("q_stride_t", C.c_int64), ("q_stride_h", C.c_int64),
        ("k_stride_b", C.c_int64), ("k_stride_t", C.c_int64), ("k_stride_h", C.c_int64),
        ("v_stride_b", C.c_int64), ("v_stride_t", C.c_int64), ("v_stride_h", C.c_int64),
        ("o_stride_b", C.c_int64), ("o_stride_t", C.c_int64), ("o_stride_h", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
        ("scale", C.c_float),
        ("key_index", C.c_void_p), ("mask_bits_q", C.c_void_p), ("mask_bits_k", C.c_void_p),
        ("tile_map", C.c_void_p),
        ("dropout_p", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("lse", C.c_void_p),
        ("dout", C.c_void_p), ("do_stride_b", C.c_int64), ("do_stride_t", C.c_int64), ("do_stride_h", C.c_int64),
        ("delta", C.c_void_p),
        ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("dq_stride_b", C.c_int64), ("dq_stride_t", C.c_int64), ("dq_stride_h", C.c_int64),
        ("dk_stride_b", C.c_int64), ("dk_stride_t", C.c_int64), ("dk_stride_h", C.c_int64),
        ("dv_stride_b", C.c_int64), ("dv_stride_t", C.c_int64), ("dv_stride_h", C.c_int64),
    ]


class MaskRule(C.Structure):
    _fields_ = [("K", C.c_int32), ("num_A", C.c_int32), ("num_B", C.c_int32), ("num_obs", C.c_int32),
                ("action_pred_steps", C.c_int32), ("atten_goal", C.c_int32), ("atten_goal_state", C.c_int32),
                ("atten_only_obs", C.c_int32), ("attn_robot_proprio_state", C.c_int32), ("n_drop", C.c_int32)]


class TokenSrc(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride_b", C.c_int64), ("stride_s", C.c_int64), ("tok_begin", C.c_int32),
                ("tok_count", C.c_int32)]


class DitSampleParams(C.Structure):
    _fields_ = [("blocks", C.c_void_p),
                ("xemb_w", C.c_void_p), ("xemb_b", C.c_void_p), ("final_w", C.c_void_p), ("final_b", C.c_void_p),
                ("pos", C.c_void_p), ("cond", C.c_void_p),
                ("coef", C.c_void_p), ("noise", C.c_void_p), ("out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("cfg_scale", C.c_float), ("ln_eps", C.c_float),
                ("depth", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("channels", C.c_int32),
                ("tokens", C.c_int32), ("bs", C.c_int32), ("steps", C.c_int32), ("reserved", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride_b", C.c_int64), ("stride_t", C.c_int64), ("T", C.c_int32)]


# every symbol include/dvla.h declares: (name, restype, argtypes)
_P, _I64, _I32, _F, _U32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint32
SYMBOLS = {
    "dvla_abi_version": (C.c_int, []),
    "dvla_gemm_bf16": (C.c_int, [C.POINTER(GemmParams), _P]),
    "dvla_gemm_ksum_partial_rows": (C.c_int64, [C.c_int32]),
    "dvla_set_gemm_variant": (None, [C.c_int]),
    "dvla_last_gemm_variant": (C.c_int, []),
    "dvla_set_gemm_schedule": (None, [C.c_int, C.c_int]),
    "dvla_get_gemm_schedule": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dvla_layernorm_fwd": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _I64, _I64, _F, _P]),
    "dvla_layernorm_bwd": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "dvla_layernorm_bwd_add": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I64, _P]),
    "dvla_layernorm_bwd_partial_rows": (_I64, []),
    "dvla_layernorm_fwd_rows": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _I64, _I64, _F, _I32, _I32, _I32, _I32, _P]),
    "dvla_layernorm_bwd_rows": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _P, _P, _I32, _P, _I64, _I64, _I32, _I32, _I32, _I32, _P]),
    "dvla_attn_fwd": (C.c_int, [C.POINTER(AttnParams), _P]),
    "dvla_attn_bwd": (C.c_int, [C.POINTER(AttnParams), _P]),
    "dvla_attn_small_fwd": (C.c_int, [C.POINTER(AttnParams), _I32, _P]),
    "dvla_attn_small_bwd": (C.c_int, [C.POINTER(AttnParams), _I32, _P]),
    "dvla_colsum": (C.c_int, [_P, _I64, _I64, _I64, _P, _P, _P]),
    "dvla_colsum_dt": (C.c_int, [_P, _I64, _I64, _I64, _P, _I32, _P, _P]),
    "dvla_colsum_partial_rows": (_I64, []),
    "dvla_dropout": (C.c_int, [_P, _P, _I64, _I64, _F, _U32, _U32, _P]),
    "dvla_act_bwd": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _F, _U32, _U32, _P]),
    "dvla_act_bwd_colsum": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _F, _U32, _U32, _P, _I32, _P, _P]),
    "dvla_act_fwd": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "dvla_ddim_cfg_step": (C.c_int, [_P, _I64, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _P]),
    "dvla_dit_sample_workspace_bytes": (C.c_int64, [_I32]),
    "dvla_dit_sample": (C.c_int, [C.POINTER(DitSampleParams), _P]),
    "dvla_dit_sample_set_stamps": (None, [_P]),
    "dvla_dit_sample_inject_timeouts": (None, [_I32]),
    "dvla_cast_f32_to_bf16": (C.c_int, [_P, _P, _I64, _P]),
    "dvla_cast_bf16_to_f32": (C.c_int, [_P, _P, _I64, _P]),
    "dvla_add": (C.c_int, [_P, _P, _P, _I64, _I64, _P]),
    "dvla_assemble_tokens": (C.c_int, [C.POINTER(TokenSrc), _I32, _P, _I64, _P, _I32, _I32, _I32, _I32, _P]),
    "dvla_mask_tables": (C.c_int, [C.POINTER(MaskRule), _P, _P, _P, _P, _P, _P]),
    "dvla_image_preprocess": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "dvla_loss_partial_len": (C.c_int64, []),
    "dvla_patch_mse_fwd": (C.c_int, [C.POINTER(FrameView), C.POINTER(FrameView), _P, _I64, _P, _P, _P]),
    "dvla_patch_mse_bwd": (C.c_int, [C.POINTER(FrameView), C.POINTER(FrameView), _P, _I64, _P, C.POINTER(FrameView), _P]),
    "dvla_cosine_loss_fwd": (C.c_int, [C.POINTER(FrameView), C.POINTER(FrameView), _I32, _I32, _I64, _P, _P, _P]),
    "dvla_cosine_loss_bwd": (C.c_int, [C.POINTER(FrameView), C.POINTER(FrameView), _I32, _I32, _I64, _P, C.POINTER(FrameView), _P]),
    "dvla_silog_loss_fwd": (C.c_int, [C.POINTER(FrameView), C.POINTER(FrameView), _I64, _F, _P, _P, _P]),
    "dvla_silog_loss_bwd": (C.c_int, [C.POINTER(FrameView), C.POINTER(FrameView), _I64, _F, _P, _P, C.POINTER(FrameView), _P]),
    "dvla_sumsq_partial_len": (C.c_int64, []),
    "dvla_sumsq_bf16": (C.c_int, [_P, _I64, _P, _P, C.c_int32, _P]),
    "dvla_adamw_bf16": (C.c_int, [_P, _P, _P, _P, _I64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I64, _P,
                                  C.c_float, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises DvlaError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvlaError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950) from the repo root. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    try:
        lib.dvla_abi_version.restype = C.c_int
        got = lib.dvla_abi_version()
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise DvlaError(f"{LIB_PATH} reports ABI version {got}, this package binds version {ABI_VERSION}: the library is stale. "
                        f"Rebuild it (`python -c 'import __graft_entry__ as g; g.build(force=True)'`).")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


CMP_LIB_PATH = os.path.join(_HERE, "libdvla_cmp.so")
_cmp = None


def load_comparator():
    """libdvla_cmp.so (include/dvla_cmp.h): the hipBLASLt yardstick.  Test / measurement infrastructure only -- no module
    of the product path calls this."""
    global _cmp
    if _cmp is None:
        if not os.path.exists(CMP_LIB_PATH):
            raise DvlaError(f"{CMP_LIB_PATH} not found (built by __graft_entry__.build())")
        lib = C.CDLL(CMP_LIB_PATH)
        lib.dvla_gemm_library_bf16.restype = C.c_int
        lib.dvla_gemm_library_bf16.argtypes = [C.POINTER(GemmParams), _P, _I64, _P]
        _cmp = lib
    return _cmp


def check(rc, what):
    if rc != 0:
        raise DvlaError(f"{what} failed: {DVLA_ERRORS.get(rc, 'error')} (code {rc})")
