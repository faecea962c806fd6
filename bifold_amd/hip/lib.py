"""ctypes binding of libbifold_hip.so (the C ABI declared in include/bifold_hip.h).

There is deliberately NO fallback: if the shared library is missing or does not
export a declared symbol, importing this module raises.  The product path runs on
hand-written HIP kernels or not at all.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

# PyTorch-ROCm ships its own libamdhip64; it must be in the process BEFORE libbifold_hip.so so
# that both resolve to ONE HIP runtime (device pointers and streams cross this boundary).
import torch  # noqa: F401

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_QUICK_GELU, ACT_ERF_GELU, ACT_ERF_GELU_GRAD, ACT_TANH_GELU, ACT_TANH_GELU_GRAD = 0, 1, 2, 3, 4, 5
ACT_ERF_GELU_SAVEGRAD, ACT_MUL_AUX = 6, 7
MASK_NONE, MASK_CAUSAL, MASK_QUERY_ROW = 0, 1, 2
TARGETS_FIRST, TARGETS_GMM, TARGETS_FIT, MAX_TARGET_POINTS = 0, 1, 2, 16
ABI_VERSION = 12

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "BIFOLD_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libbifold_hip.so"))


class LoraSlots(C.Structure):
    """Mirror of `struct bf_lora_slots` (include/bifold_hip.h)."""
    _fields_ = [("n_slots", c_int), ("rank", c_int), ("col_off", c_int * 3), ("p", c_float),
                ("seed", C.c_ulonglong * 3)]


class GemmArgs(C.Structure):
    """Mirror of `struct bf_gemm_args` (include/bifold_hip.h)."""
    _fields_ = [
        ("a", c_void_p), ("b", c_void_p), ("c", c_void_p),
        ("m", c_int), ("n", c_int), ("k", c_int),
        ("lda", c_int), ("ldb", c_int), ("ldc", c_int),
        ("cdt", c_int), ("c_dtype", c_int),
        ("trans_a", c_int), ("b_kn", c_int),
        ("alpha", c_float),
        ("bias", c_void_p),
        ("act", c_int),
        ("pre", c_void_p),
        ("aux", c_void_p), ("ld_aux", c_int),
        ("residual", c_void_p), ("ld_res", c_int), ("resid_mod", c_int),
        ("rows_in", c_int), ("rows_out", c_int), ("row_off", c_int),
        ("accumulate", c_int),
        ("c2", c_void_p),
        ("split_k", c_int),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("colsum_out", c_void_p),
        ("lr_u", c_void_p), ("lr_w", c_void_p), ("lr_k", c_int),
    ]


class GemmTnProblem(C.Structure):
    """Mirror of `struct bf_gemm_tn_problem` (include/bifold_hip.h)."""
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("c", c_void_p),
                ("m", c_int), ("n", c_int), ("ld_dy", c_int), ("ld_x", c_int), ("ldc", c_int)]


# name -> (restype, argtypes); every symbol of include/bifold_hip.h
PROTOTYPES = {
    "bf_abi_version": (c_int, []),
    "bf_last_error": (c_char_p, []),
    "bf_last_hip_error": (c_int, []),
    "bf_layernorm_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                 c_void_p, c_int, c_int, c_float, c_void_p]),
    "bf_add_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_int, c_int, c_float, c_void_p]),
    "bf_add2_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "bf_layernorm_bwd_parts": (c_int, [c_int]),
    "bf_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_void_p]),
    "bf_gemm": (c_int, [POINTER(GemmArgs), c_void_p]),
    "bf_gemm_tn_grouped_workspace_bytes": (c_size_t, [POINTER(GemmTnProblem), c_int, c_int, c_int]),
    "bf_gemm_tn_grouped": (c_int, [POINTER(GemmTnProblem), c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "bf_layernorm_fwd_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "bf_quantize_rows_fp8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "bf_gemm_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_void_p]),
    "bf_colsum": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_colsum_multi": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bf_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_float, c_int, c_int, c_void_p]),
    "bf_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "bf_attention_dropout_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                         c_int, c_int, c_float, C.c_ulonglong, c_void_p]),
    "bf_attention_dropout_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_float, C.c_ulonglong,
                                         c_void_p]),
    "bf_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_patchify_ld": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_vision_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_float, c_void_p]),
    "bf_vision_lnpre_ln1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "bf_embed_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "bf_fusion_assemble": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_void_p]),
    "bf_fusion_assemble_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_segment_assemble": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p]),
    "bf_segment_assemble_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_int, c_void_p]),
    "bf_decoder_collapse_fwd": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p]),
    "bf_decoder_collapse_bwd": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p,
                                        c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                        c_void_p]),
    "bf_decoder_project": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_void_p]),
    "bf_decoder_upsample_sigmoid": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bf_decoder_upsample_sigmoid_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                c_void_p]),
    "bf_bce_parts": (c_int, [c_int]),
    "bf_bce_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long,
                           c_long, c_void_p]),
    "bf_bce_finish": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "bf_bce_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "bf_decoder_project_bwd_ws": (c_size_t, [c_int, c_int, c_int]),
    "bf_decoder_project_bwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_mask_mul_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "bf_mask_mul_bwd": (c_int, [POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t,
                                c_int, c_void_p]),
    "bf_rows_accumulate": (c_int, [c_void_p, c_long, c_void_p, c_int, c_long, c_int, c_long, c_void_p]),
    "bf_dice_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bf_dice_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bf_focal_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_float,
                             c_void_p]),
    "bf_focal_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_void_p]),
    "bf_scale_check": (c_int, [c_void_p, c_size_t, c_float, c_void_p, c_void_p]),
    "bf_depth_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bf_depth_standardize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_scale_check_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "bf_gradscale_update": (c_int, [c_void_p, c_void_p, c_float, c_float, c_int, c_float, c_float, c_void_p]),
    "bf_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_float,
                             c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_float,
                             c_void_p]),
    "bf_sumsq": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "bf_clip_coef": (c_int, [c_void_p, c_float, c_void_p, c_void_p]),
    "bf_clip_grad_norm": (c_int, [c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "bf_debug_spin": (c_int, [c_int, C.c_longlong, c_void_p]),
    "bf_gemm_set_variant": (c_int, [c_int]),
    "bf_gemm_last_variant": (c_int, []),
    "bf_gemm_tail_workspace_bytes": (c_size_t, [POINTER(GemmArgs)]),
    "bf_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_size_t, c_void_p]),
    "bf_dropout": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_float, C.c_ulonglong, c_int, c_void_p]),
    "bf_dropout_mask": (c_int, [c_void_p, c_size_t, c_float, C.c_ulonglong, c_void_p]),
    "bf_dropout_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, C.c_ulonglong,
                                c_void_p]),
    "bf_lora_down": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, POINTER(LoraSlots), c_float, c_int,
                             c_void_p]),
    "bf_gemm_lowrank_ok": (c_int, [POINTER(GemmArgs)]),
    "bf_struct_size": (c_size_t, [c_int]),
    "bf_lora_up": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(LoraSlots), c_float, c_int, c_void_p]),
    "bf_lora_tn_workspace_elems": (c_size_t, [c_int, c_int, POINTER(LoraSlots)]),
    "bf_lora_tn": (c_int, [c_void_p, c_int, c_void_p, POINTER(c_void_p), c_int, c_void_p, c_size_t, c_int, c_int,
                           POINTER(LoraSlots), c_float, c_int, c_void_p]),
    "bf_lora_bwd_input": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, POINTER(c_void_p), c_void_p, c_size_t,
                                  c_int, c_int, POINTER(LoraSlots), c_float, c_int, c_void_p]),
    "bf_preprocess_rgb": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float),
                                  c_void_p]),
    "bf_resize_normalize_rgb": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        POINTER(c_float), POINTER(c_float), c_void_p]),
    "bf_affine_nearest": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bf_heatmap_pick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bf_gaussian_targets": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_int,
                                    c_void_p]),
}


class HipLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryError(
            f"libbifold_hip.so not found at {LIB_PATH}. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C bifold_amd/csrc`. "
            f"There is no CPU fallback for the BiFold hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.bf_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {got}, binding {ABI_VERSION}")
    return lib


lib = _load()


class HipKernelError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.bf_last_error().decode("utf-8", "replace")
        raise HipKernelError(f"{what or 'libbifold_hip'} failed (code {rc}): {msg}")
