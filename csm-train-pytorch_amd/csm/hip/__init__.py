"""ctypes binding of libcsm_hip.so (C ABI: include/csm_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback behind these calls.  Importing this module
without the built library raises; calling an op without a gfx950 device raises ``CsmHipError``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSM_HIP_LIB") or os.path.join(_HERE, "libcsm_hip.so")   # override: kernel experiments only


class CsmHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` (or `make -C csm-train-pytorch_amd/csrc`). "
        "The CSM MI355X path has no fallback implementation."
    )

lib = C.CDLL(LIB_PATH)

_p, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

_SIGS = {
    "csm_abi_version": ([], _i),
    "csm_last_error": ([], C.c_char_p),
    "csm_device_check": ([_i], _i),
    "csm_gemm_bf16": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _ll, _ll, _ll, _ll, _p], _i),
    "csm_gemm_bf16_ex": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _ll, _ll, _ll, _ll, _i, _p, _p, _i, _p], _i),
    "csm_gemm_bf16_rope": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p], _i),
    "csm_gemm_bf16_kext": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p], _i),
    "csm_skinny_nt_bf16": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "csm_gemm_bf16_dgrad_wgrad": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _f, _p], _i),
    "csm_gemm_bf16_two_wgrad": ([_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "csm_gemm_bf16_multi_wgrad": ([_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p], _i),
    "csm_set_gemm_variant": ([_i], _i),
    "csm_set_gemm256_persistent": ([_i], _i),
    "csm_get_gemm256_persistent": ([], _i),
    "csm_set_gemm_tuning": ([_i, _i], _i),
    "csm_gemm_last_kernel": ([], C.c_char_p),
    "csm_attn_bwd_workspace_bytes": ([_i, _i, _i], _ll),
    "csm_rmsnorm_fwd": ([_p, _p, _p, _p, _i, _i, _f, _p], _i),
    "csm_rmsnorm_bwd_blocks": ([], _i),
    "csm_rmsnorm_bwd": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _p], _i),
    "csm_colsum_bf16": ([_p, _i, _i, _p, _i, _p], _i),
    "csm_colsum_bf16_multi": ([_i, _p, _p, _i, _i, _i, _p], _i),
    "csm_dropout_bf16": ([_p, _i, _p, _i, _ll, _i, _f, C.c_ulonglong, _i, _p], _i),
    "csm_bias_add_bf16": ([_p, _i, _p, _ll, _i, _p], _i),
    "csm_colsum_rows_bf16": ([_p, _i, _ll, _i, _p, _i, _p], _i),
    "csm_rope": ([_p, _p, _p, _ll, _i, _i, _i, _i, _i, _p], _i),
    "csm_set_attn_variant": ([_i], _i),
    "csm_attn_last_dkv_kernel": ([], _i),
    "csm_attn64_set_debug": ([_p], _i),
    "csm_set_decode_tuning": ([_i, _i], _i),
    "csm_attn_fwd": ([_p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "csm_attn_bwd": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "csm_attn_bwd_rope": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "csm_swiglu_fwd": ([_p, _p, _ll, _i, _p], _i),
    "csm_swiglu_bwd": ([_p, _p, _p, _ll, _i, _p], _i),
    "csm_embed_fwd": ([_p, _p, _p, _p, _p, _ll, _i, _i, _i, _p], _i),
    "csm_embed_bwd_sorted": ([_p, _p, _ll, _p, _p, _ll, _p, _p, _ll, _ll, _i, _p], _i),
    "csm_rows_add_bf16": ([_p, _p, _p, _ll, _i, _i, _p], _i),
    "csm_rows_take_bf16": ([_p, _p, _p, _ll, _i, _p], _i),
    "csm_decoder_input_fwd": ([_p, _p, _p, _p, _p, _ll, _i, _i, _i, _p], _i),
    "csm_ce_fwd_bwd": ([_p, _p, _p, _p, _ll, _i, _i, _i, _f, _p], _i),
    "csm_reduce_sum_f32": ([_p, _ll, _f, _p, _p], _i),
    "csm_sumsq_blocks": ([], _i),
    "csm_sumsq_bf16": ([_p, _ll, _p, _p], _i),
    "csm_clip_coef": ([_p, _i, _f, _p, _p], _i),
    "csm_adamw_step": ([_p, _p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _i, _p, _f, _i, _p], _i),
    "csm_adamw_step_split": ([_p, _p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _i, _p, _f, _i, _p], _i),
    "csm_set_adamw_blocks": ([_i], _i),
    "csm_gemv_bf16": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_gemv_t_bf16": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_gemv_bf16_ex": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _f, _i, _p, _i, _p], _i),
    "csm_attn_decode_rope": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_gemv_attn_bf16": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_gemv_attn_at_bf16": ([_p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_attn_decode_rope_at": ([_p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_kv_append": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_attn_decode": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_sample_topk": ([_p, _p, _p, _i, _i, _i, _i, _f, _p], _i),
    "csm_rvq_encode": ([_p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "csm_conv1d_f32": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_conv_transpose1d_f32": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "csm_layernorm_f32": ([_p, _p, _p, _p, _i, _i, _f, _p], _i),
    "csm_linear_f32": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "csm_rope_half_f32": ([_p, _i, _i, _i, _f, _i, _p], _i),
    "csm_attn_window_f32": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "csm_transpose_f32": ([_p, _p, _i, _i, _p], _i),
    "csm_rvq_decode": ([_p, _p, _p, _i, _i, _i, _i, _p], _i),
}

for _name, (_args, _res) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.argtypes = _args
    _fn.restype = _res

EXPORTS = tuple(_SIGS)
if os.environ.get("CSM_ATTN_VARIANT"):                  # kernel A/B only (tools/probes): csm_set_attn_variant word
    lib.csm_set_attn_variant(int(os.environ["CSM_ATTN_VARIANT"], 0))
if os.environ.get("CSM_DECODE_TUNING"):                 # "reg,nt,rpw,regn" e.g. "1,1,1,0": kernel A/B only (tools/probes)
    for _k, _v in enumerate(os.environ["CSM_DECODE_TUNING"].split(",")):
        lib.csm_set_decode_tuning(_k, int(_v))
if os.environ.get("CSM_GEMM256_PERSISTENT") == "0":     # kernel A/B only (tools/probes)
    lib.csm_set_gemm256_persistent(0)
if os.environ.get("CSM_GEMM_TOUCH") == "0":             # kernel A/B only (tools/probes)
    lib.csm_set_gemm_tuning(0, 0)
if os.environ.get("CSM_GEMM_W4") == "0":                # kernel A/B only (tools/probes)
    lib.csm_set_gemm_tuning(1, 0)
if os.environ.get("CSM_GEMM_FAST_EPI") == "0":          # kernel A/B only (tools/probes)
    lib.csm_set_gemm_tuning(6, 0)
if os.environ.get("CSM_GEMM_W4_N6") == "0":             # kernel A/B only: 256 x 192 tiles off
    lib.csm_set_gemm_tuning(8, 0)
if os.environ.get("CSM_GEMM_W4_KEXT") == "0":           # kernel A/B only: K-extension (LoRA) products on the eight-wave kernel
    lib.csm_set_gemm_tuning(7, 0)
if os.environ.get("CSM_GEMM_STAGGER"):                  # "groups,fwd,bwd,other" (10 ns ticks): kernel A/B only (tools/probes)
    for _k, _v in enumerate(os.environ["CSM_GEMM_STAGGER"].split(",")):
        lib.csm_set_gemm_tuning(2 + _k, int(_v))


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.csm_last_error().decode("utf-8", "replace")
        raise CsmHipError(f"{what or 'libcsm_hip'} failed (code {rc}): {msg}")


def require_device(device_index: int = 0) -> None:
    """Raise unless a gfx950 GPU is visible - the product path never degrades to a CPU implementation."""
    check(lib.csm_device_check(int(device_index)), "csm_device_check")
