"""ctypes binding of libgtav_amd.so (the C-ABI declared in include/gtav_amd.h).

There is NO fallback: if the shared library is missing or a symbol is absent, import of this module
raises.  `build()` compiles it in-tree with hipcc for gfx950 (a GPU is not needed to build).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgtav_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gtav_amd.h")


class GtavError(RuntimeError):
    pass


class GtavRangeSwitch(GtavError):
    """check() under range_policy="auto": fp16 stores saturated, the layer groups in `.groups` were moved to bf16 operands; recompute."""

    def __init__(self, msg, groups=()):
        super().__init__(msg)
        self.groups = list(groups)


def build(force: bool = False) -> str:
    """Compile csrc/*.hip into libgtav_amd.so (hipcc --offload-arch=gfx950). Returns the library path."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + [HEADER_PATH]
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
            return LIB_PATH
    subprocess.run(["bash", os.path.join(src_dir, "build.sh")], check=True)
    return LIB_PATH


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_h", "input_w", "patch_size", "in_channels", "hidden_size", "depth", "num_heads")] + [
        ("mlp_ratio", C.c_float), ("external_cond_dim", C.c_int32), ("max_frames", C.c_int32), ("max_batch", C.c_int32),
        ("max_cond_rows", C.c_int32)]


class VaeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("latent_dim", "input_height", "input_width", "patch_size", "enc_dim", "enc_depth",
                                          "enc_heads", "dec_dim", "dec_depth", "dec_heads")] + [
        ("mlp_ratio", C.c_float), ("use_variational", C.c_int32), ("max_frames_per_call", C.c_int32)]


_p, _i, _f, _l = C.c_void_p, C.c_int32, C.c_float, C.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "gtav_last_error": [],
    "gtav_abi_version": [],
    "gtav_dit_create": [C.POINTER(DitConfig), C.POINTER(_p)],
    "gtav_dit_destroy": [_p],
    "gtav_dit_set_weight": [_p, C.c_char_p, _p, _l, _p],
    "gtav_dit_finalize": [_p, _p],
    "gtav_dit_get_weight": [_p, C.c_char_p, _p, _l, _p],
    "gtav_dit_forward": [_p, _p, _p, _p, _p, _i, _i, _p],
    "gtav_dit_set_schedule": [_p, C.POINTER(C.c_float), _i],
    "gtav_dit_denoise_step": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p],
    "gtav_dit_prepare_frame": [_p, _i, _i, _i, _i, _i, C.POINTER(C.c_int32), _i, _p, _p],
    "gtav_dit_check": [_p, _p],
    "gtav_dit_operand_groups": [_p, C.POINTER(C.c_int32)],
    "gtav_dit_set_operand_dtype": [_p, _i, _i],
    "gtav_dit_get_operand_dtype": [_p, _i, C.POINTER(C.c_int32)],
    "gtav_dit_autorange": [_p, C.POINTER(C.c_int32), _p],
    "gtav_dit_train_param_count": [_p, C.POINTER(C.c_int64)],
    "gtav_dit_train_enable": [_p, _p, _l],
    "gtav_dit_set_loss_scale": [_p, _f],
    "gtav_dit_set_grad_divisor": [_p, _f],
    "gtav_dit_zero_grad": [_p, _p],
    "gtav_dit_train_forward": [_p, _p, _p, _p, _p, _i, _i, _p],
    "gtav_dit_train_backward": [_p, _p, _p, _p],
    "gtav_dit_train_backward_phases": [_p, _p, _p, _i, _i, _p],
    "gtav_dit_train_param_range": [_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "gtav_dit_get_grad": [_p, C.c_char_p, _p, _l, _p],
    "gtav_dit_train_get_residual": [_p, _i, _p, _l, _p],
    "gtav_dit_adamw_step": [_p, _f, _f, _f, _f, _f, _f, _p],
    "gtav_dit_train_stats": [_p, C.POINTER(C.c_float), _p],
    "gtav_dit_get_opt_state": [_p, C.c_char_p, _p, _p, _l, _p],
    "gtav_dit_set_opt_state": [_p, C.c_char_p, _p, _p, _l, _p],
    "gtav_dit_get_opt_step": [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _p],
    "gtav_dit_set_opt_step": [_p, _l, _l, _p],
    "gtav_dit_set_graph": [_p, _i],
    "gtav_dit_set_fused_temporal": [_p, _i],
    "gtav_dit_set_fused_spatial": [_p, _i],
    "gtav_dit_fused_launches": [_p, _i, _i, _i],
    "gtav_dit_set_weight_prefetch": [_p, _i],
    "gtav_dit_profile": [_p, _i],
    "gtav_dit_profile_read": [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "gtav_timer_calibrate": [_i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), _p],
    "gtav_comm_unique_id": [_p],
    "gtav_comm_init": [C.POINTER(_p), _i, _i, _p],
    "gtav_comm_allreduce_f32": [_p, _p, _l, _i, _p],
    "gtav_comm_allgather": [_p, _p, _p, _l, _p],
    "gtav_comm_destroy": [_p],
    "gtav_vae_create": [C.POINTER(VaeConfig), C.POINTER(_p)],
    "gtav_vae_destroy": [_p],
    "gtav_vae_set_weight": [_p, C.c_char_p, _p, _l, _p],
    "gtav_vae_finalize": [_p, _p],
    "gtav_vae_get_weight": [_p, C.c_char_p, _p, _l, _p],
    "gtav_vae_encode": [_p, _p, _f, _f, _p, _i, _p],
    "gtav_vae_decode": [_p, _p, _f, _p, _f, _f, _i, _p],
    "gtav_vae_check": [_p, _p],
    "gtav_vae_set_operand_dtype": [_p, _i],
    "gtav_vae_get_operand_dtype": [_p, C.POINTER(C.c_int32)],
    "gtav_vae_profile": [_p, _i],
    "gtav_vae_profile_read": [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "gtav_clamp_frames": [_p, _i, _i, _i, _i, _f, _f, _p],
    "gtav_ddim_update": [_p, _p, _p, _i, _i, _p, _p, _i, _p],
    "gtav_add_noise": [_p, _p, _p, _p, _i, _i, _f, _p],
    "gtav_vtarget": [_p, _p, _p, _p, _i, _i, _f, _p],
    "gtav_mse": [_p, _l, _p, _l, _i, _i, _p, _p],
    "gtav_axpy_f32": [_p, _p, _f, _l, _p],
    "gtav_frames_to_u8": [_p, _p, _i, _i, _i, _p],
    "gtav_moments_to_latents": [_p, _p, _i, _i, _i, _i, _f, _p],
    "gtav_latents_to_tokens": [_p, _p, _i, _i, _i, _p],
    "gtav_strip_to_frames": [_p, _i, _i, _i, _p, _i, _i, _p],
    "gtav_resize_frames": [_p, _p, _i, _i, _i, _i, _i, _p],
    "gtav_op_gemm_f16": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p],
    "gtav_op_gemm_qkv": [_p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "gtav_op_rope_interleave": [_p, _p, _p, _i, _p],
    "gtav_op_skinny_f32": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "gtav_op_ln_modulate": [_p, _p, _i, _i, _p, _p, _i, _i, _p],
    "gtav_op_ln_affine": [_p, _p, _i, _i, _p, _p, _p],
    "gtav_op_attn_spatial": [_p, _p, _p, _p, _i, _i, _i, _p],
    "gtav_op_attn_temporal": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "gtav_op_qkv_head_major": [_p, _p, _i, _p],
    "gtav_op_gemm_tn": [_p, _p, _i, _i, _i, _p, _i, _p],
    "gtav_op_gemm_dw_grouped": [_i, _p, _p, _p, _p, _p, _p, _i, _p],
    "gtav_op_attn_spatial_bwd": [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p],
    "gtav_op_gemm_qkvt_attn": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "gtav_op_qkv_head_major_spatial": [_p, _p, _i, _p],
    "gtav_op_gemm_qkvs_attn": [_p, _p, _i, _i, _i, _p, _p, _p],
    "gtav_op_convert_f16": [_p, _i, _i, _i, _p, _i, _i, _i, _p],
    "gtav_op_gemm_splitk_ln": [_p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _i, _p],
    "gtav_op_gemm_choose_splitk": [_i, _i, _i],
    "gtav_op_gemm_resid_inplace": [_i, _i, _i],
    "gtav_op_gemm_set_stages": [_i],
    "gtav_op_gemm_set_wm": [_i],
}
_RESTYPES = {"gtav_last_error": C.c_char_p, "gtav_dit_destroy": None, "gtav_vae_destroy": None, "gtav_op_gemm_set_stages": None,
             "gtav_op_gemm_set_wm": None}

_lib = None


def load() -> C.CDLL:
    """Loads the shared library (once) and binds every symbol of SIGNATURES; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GtavError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(hipcc --offload-arch=gfx950). gtav_amd has no CPU fallback.")
    # torch's wheel bundles its own libamdhip64: it must be in the process BEFORE this library's dependency on libamdhip64 is resolved, or the process ends
    # up with two HIP runtimes — torch's sees the GPU, the one this library was bound to reports "no ROCm-capable device" (seen as build(); smoke() in one process)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


EXP_LIB_PATH = os.environ.get("GTAV_EXP_LIB") or os.path.join(_HERE, "libgtav_amd_exp.so")    # the override: A/B of two experiment builds (tools/ only)
# entry points of the experiments build only (csrc/experiments.h): the LayerNorm fold
EXP_SIGNATURES = {
    "gtav_dit_set_fold": [_p, _i, _i, _i],
    "gtav_op_gemm_fold_producer": [_p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p],
    "gtav_op_gemm_fold_consumer": [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _i, _p],
}


def load_experiments(build_if_missing: bool = True) -> C.CDLL:
    """tools/ only: the -DGTAV_EXPERIMENTS build (csrc/build.sh exp), which adds gtav_op_gemm_set_debug (timing runs with
    skipped fills / MFMAs: WRONG results) and reads the GTAV_* environment knobs.  It becomes the library every gtav_amd
    class of this process uses, so one process never mixes the two builds.  Never imported by the product or the tests."""
    global _lib
    if _lib is not None and getattr(_lib, "_gtav_experiments", False):
        return _lib
    if _lib is not None:
        raise GtavError("load_experiments() must be called before anything loaded the product library")
    if build_if_missing and not os.path.exists(EXP_LIB_PATH):
        subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh"), "exp"], check=True)
    import torch  # noqa: F401   (torch's HIP runtime first: see load())
    lib = C.CDLL(EXP_LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    for name, argtypes in EXP_SIGNATURES.items():     # csrc/experiments.h
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.gtav_op_gemm_set_debug.argtypes = [_i]
    lib.gtav_op_gemm_set_debug.restype = None
    lib.gtav_op_gemm_set_stamps.argtypes = [_p, _i]
    lib.gtav_op_gemm_set_stamps.restype = None
    lib._gtav_experiments = True
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise GtavError(load().gtav_last_error().decode() or f"gtav_amd call failed with code {rc}")


def ptr(t) -> int:
    """Device/host address of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
