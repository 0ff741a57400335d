"""ctypes binding of libmfvit_hip.so (C ABI declared in include/mfvit.h).

The product path has NO CPU / eager fallback: if the HIP library is missing, cannot be loaded, or an op is asked
to run on a non-GPU tensor, a RuntimeError is raised.
"""
import ctypes
import os

import torch  # noqa: F401  MUST precede loading libmfvit_hip.so: torch bundles its own libamdhip64.so.7 / libhsa-runtime64;
#                      whichever copy of that SONAME is mapped first serves the whole process, and mixing the system
#                      runtime with torch's bundled HSA layer leaves the extension with "no ROCm-capable device"
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFVIT_LIB") or os.path.join(_HERE, "libmfvit_hip.so")   # MFVIT_LIB: experiment builds

F32, BF16, BF16X3, F16, X3F16 = 0, 1, 2, 3, 4   # X3F16: split fp16, the qkv operand of the attention core in bf16x3 mode
EPI_BIAS, EPI_BIAS_GELU, EPI_NONE, EPI_BIAS_X3F16 = 0, 1, 3, 5


class VitCfg(Structure):
    _fields_ = [("dtype", c_int), ("batch", c_int), ("img_h", c_int), ("img_w", c_int), ("dim", c_int), ("depth", c_int),
                ("heads", c_int), ("mlp_dim", c_int), ("save_for_backward", c_int), ("stop_grad_conv1", c_int),
                ("ln_eps", c_float), ("token_input", c_int), ("tokens", c_int), ("use_pos", c_int), ("act", c_int),
                ("p_embd", c_float), ("p_attn", c_float), ("p_resid", c_float), ("seed_lo", c_uint32), ("seed_hi", c_uint32),
                ("stream_share", c_int)]


class FusionCfg(Structure):
    _fields_ = [("batch", c_int), ("tokens", c_int), ("dim", c_int), ("heads", c_int), ("num_classes", c_int),
                ("eps_pre", c_float), ("eps_post", c_float)]


P = c_void_p
I = c_int
L = c_int64
F = c_float

# name -> (restype, argtypes); must list every symbol of include/mfvit.h (tests/test_boundary_cpu.py checks it)
SIGNATURES = {
    "mfvit_abi_version": (I, []),
    "mfvit_build_info": (c_char_p, []),
    "mfvit_vit_param_count": (c_size_t, [POINTER(VitCfg)]),
    "mfvit_vit_param_layout": (I, [POINTER(VitCfg), POINTER(c_int64)]),
    "mfvit_vit_shadow_bytes": (c_size_t, [POINTER(VitCfg)]),
    "mfvit_vit_prepare_shadow": (I, [POINTER(VitCfg), P, P, P]),
    "mfvit_vit_workspace_bytes": (c_size_t, [POINTER(VitCfg)]),
    "mfvit_vit_forward": (I, [POINTER(VitCfg), P, P, P, P, P, P]),
    "mfvit_vit_backward": (I, [POINTER(VitCfg), P, P, P, P, P, I, I, P]),
    "mfvit_gpt_forward": (I, [POINTER(VitCfg), P, P, P, P, P, P]),
    "mfvit_gpt_backward": (I, [POINTER(VitCfg), P, P, P, P, P, P, P]),
    "mfvit_linear_fwd": (I, [I, I, P, L, P, L, P, P, L, P, L, I, I, I, P]),
    "mfvit_linear_dgrad_act": (I, [I, P, L, P, L, P, L, P, L, I, I, I, P]),
    "mfvit_linear_wgrad": (I, [I, P, L, P, L, P, L, I, I, I, P]),
    "mfvit_linear_wgrad_ws": (I, [I, P, L, P, L, P, L, I, I, I, P, P]),
    "mfvit_linear_wgrad_pair": (I, [I, P, L, P, L, P, L, P, I, P, L, P, L, P, L, I, I, I, P]),
    "mfvit_linear_res_ln_fwd": (I, [I, P, L, P, L, P, P, L, P, P, I, P, P, F, P, P, I, I, P]),
    "mfvit_linear_dgrad_ln_bwd": (I, [I, P, L, P, L, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "mfvit_linear_res_ln_fwd_ws": (I, [I, P, L, P, L, P, P, L, P, P, I, P, P, F, P, P, I, I, P, P]),
    "mfvit_linear_dgrad_ln_bwd_ws": (I, [I, P, L, P, L, P, P, P, P, P, P, P, P, P, P, I, I, P, P]),
    "mfvit_attention_fwd": (I, [I, P, P, P, I, I, I, I, P]),
    "mfvit_attention_bwd": (I, [I, P, P, P, P, P, P, I, I, I, I, P]),
    "mfvit_attention_qkv_dtype": (I, [I, I, I]),
    "mfvit_attention_drop_fwd": (I, [I, P, P, P, I, I, I, I, F, c_uint64, c_uint32, P]),
    "mfvit_attention_drop_bwd": (I, [I, P, P, P, P, P, I, I, I, I, F, c_uint64, c_uint32, P]),
    "mfvit_dropout_mask": (I, [F, c_uint64, c_uint32, L, P, P]),
    "mfvit_layernorm_fwd": (I, [I, P, P, I, P, P, F, P, P, I, I, P]),
    "mfvit_layernorm_bwd": (I, [I, P, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "mfvit_cast_transpose": (I, [I, P, P, P, I, I, P]),
    "mfvit_head_fwd": (I, [P, L, P, P, P, L, I, I, I, I, P]),
    "mfvit_head_bwd": (I, [P, L, P, L, P, P, L, I, P, P, I, I, I, P]),
    "mfvit_cross_entropy": (I, [P, P, P, P, P, I, I, P]),
    "mfvit_fusion_param_count": (c_size_t, [POINTER(FusionCfg)]),
    "mfvit_fusion_workspace_bytes": (c_size_t, [POINTER(FusionCfg)]),
    "mfvit_fusion_forward": (I, [POINTER(FusionCfg), P, P, P, P, P, P, P, P, P, P, P, P]),
    "mfvit_bn_stats": (I, [I, P, I, I, P, P, P]),
    "mfvit_bn_combine": (I, [P, P, P, I, I, F, F, P, P, P, P, P]),
    "mfvit_bn_apply": (I, [I, P, P, P, P, P, I, P, I, I, P]),
    "mfvit_bn_bwd_sums": (I, [I, P, P, P, P, P, I, I, I, P, P, P]),
    "mfvit_bn_bwd_apply": (I, [I, P, P, P, P, P, P, I, P, P, F, P, I, I, P]),
    "mfvit_l2norm_fwd": (I, [P, P, P, I, I, F, P]),
    "mfvit_l2norm_bwd": (I, [P, P, P, P, I, I, P]),
    "mfvit_rowdot": (I, [P, P, P, L, F, I, I, P]),
    "mfvit_cross_entropy_rows": (I, [P, L, P, P, P, P, L, I, I, P]),
    "mfvit_ema_update": (I, [P, P, F, L, P]),
    "mfvit_lars_step": (I, [P, I, I, P, F, F, F, F, P]),
    "mfvit_adam_step": (I, [P, I, F, F, F, F, F, I, P]),
    "mfvit_sgd_step": (I, [P, I, F, F, F, I, P]),
    "mfvit_amp_unscale": (I, [P, I, F, P, P]),
    "mfvit_prenorm_xattn_forward": (I, [POINTER(FusionCfg), P, P, P, P, P, P]),
    "mfvit_prenorm_xattn_backward": (I, [POINTER(FusionCfg), P, P, P, P, P, P, P, P, P]),
    "mfvit_xattn_forward": (I, [POINTER(FusionCfg), P, P, P, P, P]),
    "mfvit_xattn_backward": (I, [POINTER(FusionCfg), P, P, P, P, P, P, P]),
    "mfvit_input_transform": (I, [P, P, P, I, I, I, P, P, P, P]),
    "mfvit_eval_counts": (I, [P, L, P, I, I, P, P, P, P, P]),
    "mfvit_prof_enable": (I, [I]),
    "mfvit_set_wgrad_stream": (I, [I]),
    "mfvit_set_stream_share": (I, [I]),
    "mfvit_prof_collect": (I, [POINTER(ctypes.c_double), I]),
    "mfvit_prof_collect_tags": (I, [POINTER(ctypes.c_double), I]),
    "mfvit_prof_class_name": (c_char_p, [I]),
    "mfvit_fusion_backward": (I, [POINTER(FusionCfg), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
}

_lib = None


class MfvitError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MfvitError(
            f"libmfvit_hip.so not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for the MF-ViT hot path.")
    try:
        h = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise MfvitError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h


def source_hash():
    """sha1 over the kernel sources (csrc/*.hip, *.cuh, *.h + include/mfvit.h): stamps measurements (PMC traffic files under
    profiles/) with the code they were taken on, so that bench.py never reports a figure measured on other kernels."""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cuh", ".h")))
    files.append(os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "mfvit.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def check(rc, what):
    if rc != 0:
        raise MfvitError(f"{what} failed with code {rc} (-22 = invalid argument, -5 = launch error, -38 = not built for this dtype / size)")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MfvitError("MF-ViT HIP ops need tensors on a ROCm GPU ('cuda' device); there is no CPU fallback")


PRECISIONS = ("bf16x3", "bf16", "fp16", "fp32")


def dtype_code(precision):
    """'bf16x3' = split bf16 (three bf16 MFMAs per product: f32-grade results, meets the 1e-3 logits gate); 'bf16' = throughput mode;
    'fp16' = the reference's autocast arithmetic (pair with mfvit.amp.GradScaler); 'fp32' = exact f32 MFMA."""
    if precision in ("bf16x3", "split", "split-bf16"):
        return BF16X3
    if precision in ("bf16", "bfloat16"):
        return BF16
    if precision in ("fp16", "f16", "float16", "half"):
        return F16
    if precision in ("fp32", "f32", "float32"):
        return F32
    raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
