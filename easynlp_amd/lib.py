"""ctypes binding of libezclip_hip.so (the C ABI in include/ezclip.h).

PyTorch is used for device memory and streams only: every call here passes raw
``tensor.data_ptr()`` device pointers and the current HIP stream handle.
There is NO CPU fallback: if the shared library is missing or a tensor is not
on a GPU, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# EZCLIP_LIB: load another build of the same library (experiment variants of tools/build_variants.py); unset = the in-tree one
LIB_PATH = os.environ.get("EZCLIP_LIB") or os.path.join(_HERE, "csrc", "libezclip_hip.so")

DTYPE_F32 = 0
DTYPE_BF16 = 1
ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF, ACT_TANH = 0, 1, 2, 3
OPT_TEXT_POOLER, OPT_VISION_FROZEN, OPT_TEXT_LN_EPS, OPT_TEXT_PAD_ID, OPT_BLOCK_LN_EPS, OPT_TEXT_EOT_ID = 1, 2, 3, 4, 5, 6


class EzclipError(RuntimeError):
    pass


class EzclipConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "embed_dim", "image_resolution", "vision_layers", "vision_width", "vision_patch_size",
        "vocab_size", "text_hidden_size", "text_intermediate_size", "text_max_position_embeddings",
        "text_num_attention_heads", "text_num_hidden_layers", "text_type_vocab_size", "compute_dtype")]


class EzclipRnConfig(C.Structure):
    """ezclip_rn_config (include/ezclip.h)"""
    _fields_ = [("layers", C.c_int32 * 4), ("width", C.c_int32), ("output_dim", C.c_int32), ("image_resolution", C.c_int32),
                ("compute_dtype", C.c_int32)]


class EzclipAttentionOpts(C.Structure):
    """ezclip_attention_opts (include/ezclip.h)"""
    _fields_ = [("causal", C.c_int32), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64), ("dropout_site", C.c_uint32),
                ("reserved_", C.c_uint32)]


def attention_opts(causal=False, dropout=None):
    """None, or a pointer to the options of one op-level attention call; dropout = (p, seed, site)."""
    if not causal and not dropout:
        return None
    o = EzclipAttentionOpts()
    o.causal = 1 if causal else 0
    if dropout:
        o.dropout_p, o.dropout_seed, o.dropout_site = float(dropout[0]), int(dropout[1]), int(dropout[2])
    return C.cast(C.pointer(o), C.c_void_p)


class EzclipImageDesc(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("width", C.c_int32), ("height", C.c_int32)]


_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float

# ezclip_progress_fn (include/ezclip.h): void (*)(void* user, int tower, int stage)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)
STAGE_HEAD, STAGE_EMBED = 1000000, -1


class EzclipGemmDesc(C.Structure):
    """ezclip_gemm_desc (include/ezclip.h)"""
    _fields_ = [("a", _vp), ("lda", _i64), ("b", _vp), ("ldb", _i64), ("c", _vp), ("ldc", _i64), ("c2", _vp),
                ("bias", _vp), ("residual", _vp), ("ldr", _i64), ("u", _vp), ("ldu", _i64),
                ("ln_stats", _vp), ("ln_c1", _vp), ("ln_c2", _vp), ("rowstat_part", _vp), ("colsum", _vp),
                ("alpha", _f), ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("act", C.c_int32),
                ("dtype", C.c_int32), ("out_f32", C.c_int32), ("force_kernel", C.c_int32)]

# name -> (restype, argtypes): every symbol include/ezclip.h declares
SIGNATURES = {
    "ezclip_last_error": (C.c_char_p, []),
    "ezclip_version": (C.c_char_p, []),
    "ezclip_create": (_i, [C.POINTER(EzclipConfig), C.POINTER(_vp)]),
    "ezclip_create_ex": (_i, [C.POINTER(EzclipConfig), _i, C.POINTER(_vp)]),
    "ezclip_destroy": (None, [_vp]),
    "ezclip_num_params": (_i, [_vp]),
    "ezclip_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i)]),
    "ezclip_bind_param": (_i, [_vp, C.c_char_p, _vp, _vp, C.POINTER(_i64), _i]),
    "ezclip_shadow_bytes": (_sz, [_vp, _i]),
    "ezclip_set_shadow": (_i, [_vp, _vp, _sz, _i]),
    "ezclip_refresh_weights": (_i, [_vp, _vp]),
    "ezclip_image_workspace_bytes": (_sz, [_vp, _i, _i]),
    "ezclip_text_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "ezclip_encode_image": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _i, _vp]),
    "ezclip_encode_text": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _i, _vp]),
    "ezclip_similarity": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ezclip_infonce_from_logits": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ezclip_infonce_from_logits_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ezclip_cross_entropy_diag": (_i, [_vp, _i, _i, C.c_int64, _vp, _vp, _vp]),
    "ezclip_cross_entropy_diag_bwd": (_i, [_vp, _i, _i, C.c_int64, _vp, _vp, _vp, _vp]),
    "ezclip_infonce_workspace_bytes": (_sz, [_i, _i, _i]),
    "ezclip_infonce_fused": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ezclip_infonce_tiled_workspace_bytes": (_sz, [_i, _i, _i]),
    "ezclip_rn_create": (_i, [_vp, C.POINTER(_vp)]),
    "ezclip_rn_destroy": (None, [_vp]),
    "ezclip_rn_num_params": (_i, [_vp]),
    "ezclip_rn_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(_i)]),
    "ezclip_rn_bind_param": (_i, [_vp, C.c_char_p, _vp]),
    "ezclip_rn_shadow_bytes": (_sz, [_vp]),
    "ezclip_rn_set_shadow": (_i, [_vp, _vp, _sz]),
    "ezclip_rn_refresh_weights": (_i, [_vp, _vp]),
    "ezclip_rn_workspace_bytes": (_sz, [_vp, _i]),
    "ezclip_rn_encode_image": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "ezclip_pack_text_meta": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, C.POINTER(C.c_int), _vp]),
    "ezclip_pack_text_meta_result": (_i, [_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ezclip_infonce_tiled": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ezclip_backward_image": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "ezclip_backward_text": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ezclip_recall_ranks": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ezclip_recall_ranks_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ezclip_recall_paired_scores": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ezclip_recall_ranks_fused": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ezclip_rn_train_shadow_bytes": (_sz, [_vp]),
    "ezclip_rn_set_train_shadow": (_i, [_vp, _vp, _sz]),
    "ezclip_rn_refresh_train_weights": (_i, [_vp, _vp]),
    "ezclip_rn_bind_grad": (_i, [_vp, C.c_char_p, _vp]),
    "ezclip_rn_train_saved_bytes": (_sz, [_vp, _i]),
    "ezclip_rn_train_scratch_bytes": (_sz, [_vp, _i]),
    "ezclip_rn_encode_image_train": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ezclip_rn_backward": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "ezclip_op_rn_bn_scratch_bytes": (_sz, [_i64, _i]),
    "ezclip_op_rn_bn_train_fwd": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ezclip_op_rn_bn_train_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "ezclip_op_rn_avgpool2_bwd": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "ezclip_op_rn_im2col3x3": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "ezclip_op_rn_pack_conv_dgrad": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "ezclip_op_rn_unpack_wgrad": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "ezclip_op_conv3x3_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "ezclip_debug_set": (_i, [_i, _i]),
    "ezclip_profile_begin": (_i, []),
    "ezclip_profile_end": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    "ezclip_op_gemm_nt": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "ezclip_op_gemm_nt_ex": (_i, [C.POINTER(EzclipGemmDesc), _vp]),
    "ezclip_op_layernorm_stats": (_i, [_vp, _i64, _f, _i, _i, _i, _vp, _vp]),
    "ezclip_set_backward_progress": (_i, [_vp, _vp, _vp]),
    "ezclip_backward_progress_events": (_i, [_vp, _i]),
    "ezclip_backward_progress_drain": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "ezclip_stream_wait_event": (_i, [_vp, _vp]),
    "ezclip_op_gemm_tn": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "ezclip_op_gemm_tn_conv3x3": (_i, [_vp, _i64, _vp, _i, _i, _i, _i, _vp, _i64, _i, _i, _i, _vp]),
    "ezclip_op_rn_wgrad3x3_c64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i64, _i, _vp]),
    "ezclip_op_rn_tn_skinny": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "ezclip_op_layernorm": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp]),
    "ezclip_op_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ezclip_op_attention": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ezclip_op_attention_bwd": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ezclip_op_attention_bwd_bias": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _i, _i, _i, _i, _vp, _vp]),
    "ezclip_preprocess_workspace_bytes": (_sz, [C.POINTER(EzclipImageDesc), _i, _i, _i]),
    "ezclip_preprocess_images": (_i, [_vp, C.POINTER(EzclipImageDesc), _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp,
                                      _sz, _vp]),
    "ezclip_op_resample_table": (_i, [_i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i]),
    "ezclip_op_resample_table_device": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "ezclip_set_option": (_i, [_vp, _i, C.c_double]),
    "ezclip_encode_text_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _i, _vp]),
    "ezclip_encode_text_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp]),
    "ezclip_backward_text_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ezclip_backward_text_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ezclip_set_text_dropout": (_i, [_vp, _f, _f, C.c_uint64]),
    "ezclip_op_dropout": (_i, [_vp, _vp, _vp, _i, _i, _f, C.c_uint64, C.c_uint32, _i, _vp]),
    "ezclip_op_dropout_mask": (_i, [_f, C.c_uint64, C.c_uint32, _i, _i, _vp, _vp, _vp]),
    "ezclip_op_attention_cls": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "ezclip_op_attention_cls_bwd": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "ezclip_op_cast_from_f32": (_i, [_vp, _vp, _i64, _i, _vp]),
    "ezclip_op_cast_to_f32": (_i, [_vp, _vp, _i64, _i, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EzclipError(
            "libezclip_hip.so not found at %s -- build it with `python easynlp_amd/csrc/build.py` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().ezclip_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise EzclipError("%s failed (rc=%d): %s" % (what or "ezclip call", rc, last_error()))


def stream_ptr(stream=None) -> int:
    """hipStream_t of a torch stream (default: the current one)."""
    return int((stream if stream is not None else torch.cuda.current_stream()).cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous GPU tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EzclipError("ezclip needs GPU tensors (got device %s); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise EzclipError("ezclip needs contiguous tensors")
    return int(t.data_ptr())


def torch_dtype(dtype: int) -> torch.dtype:
    return torch.bfloat16 if dtype == DTYPE_BF16 else torch.float32


def dtype_code(name) -> int:
    if name in (DTYPE_F32, DTYPE_BF16):
        return int(name)
    s = str(name).lower().replace("torch.", "")
    if s in ("bf16", "bfloat16"):
        return DTYPE_BF16
    if s in ("fp32", "f32", "float32", "float"):
        return DTYPE_F32
    raise EzclipError("unknown compute dtype %r (use 'bf16' or 'fp32')" % (name,))


def alloc_bytes(nbytes: int, device) -> torch.Tensor:
    """256-byte aligned scratch owned by the torch caching allocator."""
    t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t[off:off + int(nbytes)]


# ---------------------------------------------------------------- op wrappers (tests, microbench)

def op_gemm_nt(a: torch.Tensor, b: torch.Tensor, bias=None, residual=None, act=ACT_NONE, out_f32=False,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = act(A @ B^T + bias) + residual with A [M,K], B [N,K] (row stride = K)."""
    lib = load()
    dt = DTYPE_BF16 if a.dtype == torch.bfloat16 else DTYPE_F32
    M, K = a.shape
    N = b.shape[0]
    odt = torch.float32 if (out_f32 or dt == DTYPE_F32) else torch.bfloat16
    c = out if out is not None else torch.empty((M, N), dtype=odt, device=a.device)
    check(lib.ezclip_op_gemm_nt(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(c), c.stride(0), ptr(bias),
                                ptr(residual), residual.stride(0) if residual is not None else 0, M, N, K, act, dt,
                                1 if (out_f32 and dt == DTYPE_BF16) else 0, stream_ptr()), "op_gemm_nt")
    return c


def op_gemm_nt_ex(a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor] = None, bias=None, residual=None, u=None,
                  c2=None, ln_stats=None, ln_c1=None, ln_c2=None, rowstat_part=None, colsum=None, act=ACT_NONE,
                  alpha: float = 1.0, force_kernel: int = -1) -> torch.Tensor:
    """Every epilogue of the NT GEMM (ezclip_op_gemm_nt_ex); force_kernel: -1 heuristic, 0 128x128, 2 persistent 8-phase,
    24 8-phase with one workgroup per tile."""
    lib = load()
    dt = DTYPE_BF16 if a.dtype == torch.bfloat16 else DTYPE_F32
    M, K = a.shape
    N = b.shape[0]
    if c is None:
        c = torch.empty((M, N), dtype=a.dtype, device=a.device)
    d = EzclipGemmDesc()
    d.a, d.lda, d.b, d.ldb, d.c, d.ldc = ptr(a) if a.is_contiguous() else a.data_ptr(), a.stride(0), ptr(b), b.stride(0), ptr(c), c.stride(0)
    d.c2, d.bias = ptr(c2), ptr(bias)
    d.residual, d.ldr = ptr(residual), (residual.stride(0) if residual is not None else 0)
    d.u, d.ldu = ptr(u), (u.stride(0) if u is not None else 0)
    d.ln_stats, d.ln_c1, d.ln_c2 = ptr(ln_stats), ptr(ln_c1), ptr(ln_c2)
    d.rowstat_part, d.colsum = ptr(rowstat_part), ptr(colsum)
    d.alpha, d.m, d.n, d.k, d.act, d.dtype, d.out_f32, d.force_kernel = float(alpha), M, N, K, int(act), dt, 0, int(force_kernel)
    check(lib.ezclip_op_gemm_nt_ex(C.byref(d), stream_ptr()), "op_gemm_nt_ex")
    return c


def op_layernorm_stats(x: torch.Tensor, eps: float) -> torch.Tensor:
    """[rows, 2] float32: (rstd, -mean * rstd) of LayerNorm over the last dim."""
    lib = load()
    dt = DTYPE_BF16 if x.dtype == torch.bfloat16 else DTYPE_F32
    out = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    check(lib.ezclip_op_layernorm_stats(ptr(x), x.stride(0), float(eps), x.shape[0], x.shape[1], dt, ptr(out), stream_ptr()),
          "op_layernorm_stats")
    return out


def op_layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float, want_stats=False):
    lib = load()
    dt = DTYPE_BF16 if x.dtype == torch.bfloat16 else DTYPE_F32
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    check(lib.ezclip_op_layernorm(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(g), ptr(b), eps, rows, d, dt,
                                  ptr(mean), ptr(rstd), stream_ptr()), "op_layernorm")
    return (y, mean, rstd) if want_stats else y


def op_attention(qkv: torch.Tensor, batch: int, seq_len: int, heads: int, key_bias=None, want_lse=False, causal=False, dropout=None):
    """qkv: [batch*seq_len, 3*heads*64] packed (q | k | v); returns ctx [batch*seq_len, heads*64].  dropout = (p, seed, site)."""
    lib = load()
    dt = DTYPE_BF16 if qkv.dtype == torch.bfloat16 else DTYPE_F32
    D = heads * 64
    esz = qkv.element_size()
    ctx = torch.empty((batch * seq_len, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((batch, heads, seq_len), dtype=torch.float32, device=qkv.device) if want_lse else None
    base = ptr(qkv)
    check(lib.ezclip_op_attention(base, base + D * esz, base + 2 * D * esz, 3 * D, ptr(ctx), D, ptr(key_bias),
                                  ptr(lse), batch, seq_len, heads, dt, attention_opts(causal, dropout), stream_ptr()), "op_attention")
    return (ctx, lse) if want_lse else ctx


def op_attention_bwd(qkv: torch.Tensor, ctx: torch.Tensor, dctx: torch.Tensor, lse: torch.Tensor, batch: int, seq_len: int,
                     heads: int, key_bias=None, causal=False, dropout=None) -> torch.Tensor:
    """Gradient of op_attention w.r.t. the packed qkv (same [batch*seq_len, 3*heads*64] layout)."""
    lib = load()
    dt = DTYPE_BF16 if qkv.dtype == torch.bfloat16 else DTYPE_F32
    D = heads * 64
    esz = qkv.element_size()
    dqkv = torch.zeros_like(qkv)
    base, dbase = ptr(qkv), ptr(dqkv)
    check(lib.ezclip_op_attention_bwd(base, base + D * esz, base + 2 * D * esz, 3 * D, ptr(ctx), ptr(dctx), D,
                                      ptr(key_bias), ptr(lse), dbase, dbase + D * esz, dbase + 2 * D * esz, batch,
                                      seq_len, heads, dt, attention_opts(causal, dropout), stream_ptr()), "op_attention_bwd")
    return dqkv


def op_dropout(x: torch.Tensor, p: float, seed: int, site: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = dropout(x) [+ residual] with the library's counter-based mask (x: [rows, d])."""
    lib = load()
    dt = DTYPE_BF16 if x.dtype == torch.bfloat16 else DTYPE_F32
    y = torch.empty_like(x)
    check(lib.ezclip_op_dropout(ptr(x), ptr(residual), ptr(y), x.shape[0], x.shape[1], float(p), int(seed), int(site),
                                dt, stream_ptr()), "op_dropout")
    return y


def op_dropout_mask(p: float, seed: int, site: int, rows: int, cols: int, device, want_words=False):
    """keep[rows, cols] (uint8) of a dropout site -- and the raw Philox words (int64 holding uint32) if asked."""
    lib = load()
    keep = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    words = torch.empty((rows, cols), dtype=torch.int32, device=device) if want_words else None
    check(lib.ezclip_op_dropout_mask(float(p), int(seed), int(site), rows, cols, ptr(keep), ptr(words), stream_ptr()),
          "op_dropout_mask")
    if want_words:
        return keep, words.to(torch.int64) & 0xFFFFFFFF
    return keep


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # easynlp/appzoo/clip/data.py:101
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def pack_images(images, pin: bool = False) -> dict:
    """Decoded images -> ONE contiguous uint8 buffer + descriptors, the layout ``ezclip_preprocess_images`` reads:
    ``{"data": uint8 [total] (each image HWC RGB, starts 16-byte aligned), "desc": int64 [n, 3] (offset, width, height)}``.
    Pure host work with no GPU call: ``batch_fn`` of the drop-in datasets can run it inside the DataLoader workers
    (``pack_batches=True``), so that the per-image copies are spread over the workers and the training process only moves
    one tensor to the device.  ``images``: uint8 HWC RGB numpy arrays ([H, W] / [H, W, 1] greyscale is replicated, as
    ``convert('RGB')`` does after the reference's resize).  ``pin``: allocate the buffer in pinned memory (only in the
    process that owns the GPU context -- never in a DataLoader worker)."""
    import numpy as np
    arrs = []
    for im in images:
        a = np.asarray(im)
        if a.dtype != np.uint8:
            raise EzclipError("preprocess_images needs uint8 pixels (got %s)" % a.dtype)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.ndim != 3 or a.shape[2] not in (1, 3):
            raise EzclipError("preprocess_images needs RGB or greyscale images [H, W, 3]; got shape %s -- convert('RGB') "
                              "first (palette / alpha images resize differently in the reference)" % (a.shape,))
        if a.shape[2] == 1:
            a = np.repeat(a, 3, axis=2)
        arrs.append(np.ascontiguousarray(a))
    n = len(arrs)
    if n == 0:
        raise EzclipError("preprocess_images: empty batch")
    desc = torch.empty((n, 3), dtype=torch.int64)
    off = 0
    for i, a in enumerate(arrs):
        desc[i, 0], desc[i, 1], desc[i, 2] = off, a.shape[1], a.shape[0]
        off += (a.size + 15) // 16 * 16
    data = torch.empty(off + 16, dtype=torch.uint8)
    if pin:
        data = data.pin_memory()
    hv = data.numpy()
    for i, a in enumerate(arrs):
        o = int(desc[i, 0])
        hv[o:o + a.size] = a.reshape(-1)
    return {"data": data, "desc": desc}


def is_packed_images(x) -> bool:
    return isinstance(x, dict) and "data" in x and "desc" in x


def preprocess_images(images, size: int = 224, crop: int = 224, mean=CLIP_MEAN, std=CLIP_STD, device="cuda") -> torch.Tensor:
    """Decoded images -> float32 ``pixel_values`` [n, 3, crop, crop] on the GPU (ezclip_preprocess_images: Pillow-exact
    bicubic resize of the shorter side to ``size``, centre crop, /255, normalise).  ``images``: a list of uint8 HWC RGB
    numpy arrays (see ``pack_images``) or the dict ``pack_images`` returns (possibly pinned / already on the device)."""
    lib = load()
    packed_in = images if is_packed_images(images) else pack_images(images, pin=torch.cuda.is_available())
    data, dtab = packed_in["data"], packed_in["desc"]
    n = int(dtab.shape[0])
    rows = dtab.cpu().tolist()
    desc = (EzclipImageDesc * n)()
    for i, (o, w, h) in enumerate(rows):
        desc[i].offset, desc[i].width, desc[i].height = int(o), int(w), int(h)
    # one host -> device copy: asynchronous from pinned memory (the list path packs into a pinned buffer; a DataLoader with
    # pin_memory=True pins the workers' packed batches in its own thread), a single staged copy from pageable memory
    host = None if data.is_cuda else data
    packed = data if data.is_cuda else data.to(device, non_blocking=True)
    nbytes = lib.ezclip_preprocess_workspace_bytes(desc, n, size, crop)
    if nbytes == 0:
        raise EzclipError("preprocess_images: %s" % last_error())
    ws = alloc_bytes(nbytes, packed.device)
    out = torch.empty((n, 3, crop, crop), dtype=torch.float32, device=packed.device)
    m3, s3 = (_f * 3)(*mean), (_f * 3)(*std)
    check(lib.ezclip_preprocess_images(ptr(packed), desc, n, size, crop, m3, s3, ptr(out), ptr(ws), ws.numel(), stream_ptr()),
          "preprocess_images")
    # the pinned host copy and the workspace must outlive the enqueued work
    out._ezclip_keepalive = (host, packed, ws)
    return out


def similarity(a: torch.Tensor, b: torch.Tensor, logit_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib.ezclip_similarity(ptr(a), ptr(b), a.shape[0], b.shape[0], a.shape[1], ptr(logit_scale), ptr(out),
                                stream_ptr()), "similarity")
    return out
