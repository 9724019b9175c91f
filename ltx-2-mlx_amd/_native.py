"""ctypes binding of libltx2hip.so (C ABI declared in include/ltx2hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a call is made
without a GPU tensor, this module raises.  PyTorch is used only to own device memory and
streams; every compute call goes through the C ABI with raw device pointers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTX2HIP_LIB / LTX2HIP_LIB_F16: another build of the same library (A/B timing of kernel changes on one GPU box)
LIB_PATH = os.environ.get("LTX2HIP_LIB") or os.path.join(_HERE, "lib", "libltx2hip.so")
# the same sources compiled with -DLTX2_F16: IEEE-half activations / weights (the reference's default compute dtype)
LIB_PATH_F16 = os.environ.get("LTX2HIP_LIB_F16") or os.path.join(_HERE, "lib", "libltx2hip_f16.so")

ABI_VERSION = 3         # LTX2_ABI_VERSION of include/ltx2hip.h these signatures were written against
OK, E_INVALID, E_HIP, E_STATE = 0, -1, -2, -3
DTYPE_BF16, DTYPE_F32, DTYPE_FP8_E4M3FN = 0, 1, 2
MODEL_VIDEO_ONLY, MODEL_AUDIO_VIDEO = 0, 1
EPI_BF16, EPI_GELU_BF16, EPI_SILU_BF16, EPI_F32, EPI_RESID_GATE_F32, EPI_ADD_BF16 = range(6)
ROUTE_SKINNY, ROUTE_V4_224, ROUTE_V4_256, ROUTE_V4_W8_224, ROUTE_V4_W8_256, ROUTE_V4_F8_224, ROUTE_V4_F8_256, ROUTE_PP, ROUTE_SMALL, ROUTE_NARROW = range(10)
VAE_RES, VAE_UPSAMPLE = 0, 1
VAE_MAX_BLOCKS = 16

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class DitConfig(C.Structure):
    _fields_ = [("num_layers", i32), ("num_heads", i32), ("head_dim", i32), ("in_channels", i32),
                ("out_channels", i32), ("caption_channels", i32), ("norm_eps", f32), ("timestep_scale", f32),
                ("model_type", i32), ("audio_heads", i32), ("audio_head_dim", i32), ("audio_in_channels", i32),
                ("audio_out_channels", i32), ("cross_attention_adaln", i32), ("apply_gated_attention", i32),
                ("av_ca_timestep_scale", f32)]


class VaeConfig(C.Structure):
    _fields_ = [("n_blocks", i32), ("kind", i32 * VAE_MAX_BLOCKS), ("num_layers", i32 * VAE_MAX_BLOCKS),
                ("stride", (i32 * 3) * VAE_MAX_BLOCKS), ("multiplier", i32 * VAE_MAX_BLOCKS),
                ("residual", i32 * VAE_MAX_BLOCKS), ("base_channels", i32), ("latent_channels", i32),
                ("timestep_conditioning", i32), ("decode_noise_scale", f32)]


# name -> (restype, argtypes); every symbol of include/ltx2hip.h
SIGNATURES = {
    "ltx2_last_error": (C.c_char_p, []),
    "ltx2_clear_error": (None, []),
    "ltx2_abi_version": (i32, []),
    "ltx2_gemm_bf16": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp, vp, i64, vp]),
    "ltx2_gemm_qkv_vt": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, vp, i32, i32, i32, C.POINTER(i32), vp]),
    "ltx2_gemm_w8a16": (i32, [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp, vp]),
    "ltx2_gemm_bf16_rowss": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, vp, C.POINTER(i32), vp]),
    "ltx2_gemm_bf16_fold": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, i64, vp, vp, i64, vp, i64, i32, i32, f32, C.POINTER(i32), vp]),
    "ltx2_flash_attn_keymask": (i32, [vp, i64, vp, i64, vp, i32, vp, i64, i32, i32, i32, i32, f32, vp, vp, vp]),
    "ltx2_adaln_rmsnorm2": (i32, [vp, i64, vp, vp, i64, i32, i32, f32, vp, vp, vp, vp, vp]),
    "ltx2_flash_attn_gated": (i32, [vp, i64, vp, i64, vp, i32, vp, i64, i32, i32, i32, i32, f32, vp, i32, vp]),
    "ltx2_flash_attn_gated_parts": (i32, [vp, i64, vp, i64, vp, i32, vp, i64, i32, i32, i32, i32, f32, vp, i64, vp, vp, i32, vp, vp]),
    "ltx2_flash_attn_rowscale": (i32, [vp, i64, vp, i64, vp, i32, vp, i64, i32, i32, i32, i32, f32, vp, i32, i32, f32, vp]),
    "ltx2_gemm_route": (i32, [i32, i32, i32, i32, i32, i32]),
    "ltx2_quantize_rows_fp8": (i32, [vp, i64, i32, i32, vp, i64, vp, vp]),
    "ltx2_gemm_fp8": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp, vp]),
    "ltx2_gemm_fp8_qkv_vt": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, i32, i32, i32, C.POINTER(i32), vp]),
    "ltx2_gemv_f32": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "ltx2_conv3d_fused": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ltx2_groupnorm_silu": (i32, [vp, vp, vp, i64, i32, i32, f32, vp, vp, vp, i32, vp]),
    "ltx2_s2d_downsample": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ltx2_latent_normalize_nchw": (i32, [vp, vp, vp, vp, i32, i64, vp]),
    "ltx2_adaln_rmsnorm": (i32, [vp, i64, vp, i64, i32, i32, f32, i32, vp, vp, vp, vp, i64, vp]),
    "ltx2_adaln_rmsnorm_fp8": (i32, [vp, i64, vp, i64, vp, i64, vp, i32, i32, f32, i32, vp, vp, vp, vp, i64, vp]),
    "ltx2_qknorm_rope": (i32, [vp, i64, i32, i32, i32, i32, vp, i32, vp, f32, vp, vp, vp]),
    "ltx2_vt_transpose": (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    "ltx2_flash_attn": (i32, [vp, i64, vp, i64, vp, i32, vp, i64, i32, i32, i32, i32, f32, vp]),
    "ltx2_attn_head_gate": (i32, [vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ltx2_rope_tables": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "ltx2_timestep_sinusoid": (i32, [vp, i64, f32, i32, i32, vp, vp, vp]),
    "ltx2_dequant_fp8_e4m3fn": (i32, [vp, f32, vp, i64, vp]),
    "ltx2_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
    "ltx2_x0_from_velocity": (i32, [vp, vp, vp, i64, f32, vp, i32, i32, vp]),
    "ltx2_euler_step": (i32, [vp, vp, vp, vp, f32, f32, vp, i32, i32, vp]),
    "ltx2_vae_prepare_latent": (i32, [vp, vp, vp, vp, f32, vp, i32, i64, vp]),
    "ltx2_pixnorm_mod_silu": (i32, [vp, vp, i64, i32, f32, vp, vp, i32, i32, vp]),
    "ltx2_vae_unpatchify": (i32, [vp, vp, i32, i32, i32, vp]),
    "ltx2_video_to_uint8": (i32, [vp, vp, i32, i32, i32, vp]),
    "ltx2_video_chunk_to_uint8": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ltx2_tile_blend_accumulate": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ltx2_tile_blend_finish": (i32, [vp, vp, i64, vp]),
    "ltx2_dit_create": (i32, [C.POINTER(DitConfig), C.POINTER(vp)]),
    "ltx2_dit_destroy": (None, [vp]),
    "ltx2_dit_set_weight": (i32, [vp, C.c_char_p, vp, i32, i64]),
    "ltx2_dit_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "ltx2_dit_bind_workspace": (i32, [vp, vp, i64, i32, i32, i32]),
    "ltx2_dit_workspace_bytes_av": (i64, [vp, i32, i32, i32, i32, i32]),
    "ltx2_dit_bind_workspace_av": (i32, [vp, vp, i64, i32, i32, i32, i32, i32]),
    "ltx2_dit_prepare": (i32, [vp, vp, i32, vp, vp, vp]),
    "ltx2_dit_prepare_av": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "ltx2_dit_forward_av": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "ltx2_dit_denoise_step_av": (i32, [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp]),
    "ltx2_dit_graph_capture_av": (i32, [vp, vp, vp, C.POINTER(f32), i32, vp]),
    "ltx2_dit_forward": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "ltx2_dit_denoise_step": (i32, [vp, vp, vp, i32, vp, vp, vp, f32, f32, vp, vp]),
    "ltx2_dit_graph_capture": (i32, [vp, vp, C.POINTER(f32), i32, vp]),
    "ltx2_dit_graph_capture_cond": (i32, [vp, vp, C.POINTER(f32), i32, vp, i64, vp, i64, vp]),
    "ltx2_dit_graph_capture_cond_av": (i32, [vp, vp, vp, C.POINTER(f32), i32, vp, i64, vp, i64, vp, i64, vp, i64, vp]),
    "ltx2_dit_graph_launch": (i32, [vp, vp]),
    "ltx2_dit_set_context_mask": (i32, [vp, i32, vp, i32, vp]),
    "ltx2_dit_set_option": (i32, [vp, C.c_char_p, i32]),
    "ltx2_dit_profile_begin": (i32, [vp, i32]),
    "ltx2_dit_profile_end": (i32, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]),
    "ltx2_vae_create": (i32, [C.POINTER(VaeConfig), C.POINTER(vp)]),
    "ltx2_vae_destroy": (None, [vp]),
    "ltx2_vae_set_weight": (i32, [vp, C.c_char_p, vp, i32, i64]),
    "ltx2_vae_set_timestep_multiplier": (i32, [vp, f32]),
    "ltx2_vae_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "ltx2_vae_bind_workspace": (i32, [vp, vp, i64]),
    "ltx2_vae_decode": (i32, [vp, vp, i32, i32, i32, f32, vp, i32, vp, vp]),
    "ltx2_vae_out_frames": (i32, [vp, i32]),
}

_libs: dict = {}


class NativeLibraryMissing(RuntimeError):
    pass


def lib(dtype: Optional[torch.dtype] = None) -> C.CDLL:
    """Load the library build for `dtype` (once): bfloat16 (default / None) -> libltx2hip.so, float16 -> libltx2hip_f16.so.  Both export
    the same C ABI; dtype code DTYPE_BF16 means "the build's 16-bit type".  Fails loudly if it has not been built."""
    key = torch.float16 if dtype == torch.float16 else torch.bfloat16
    if dtype not in (None, torch.float16, torch.bfloat16):
        raise ValueError(f"no library build for compute dtype {dtype} (bfloat16 and float16 exist)")
    if key not in _libs:
        path = LIB_PATH_F16 if key == torch.float16 else LIB_PATH
        env = "LTX2HIP_LIB_F16" if key == torch.float16 else "LTX2HIP_LIB"
        if not os.path.exists(path):
            raise NativeLibraryMissing(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C ltx-2-mlx_amd/csrc`). There is no CPU fallback for the hot path.")
        l = C.CDLL(path)
        l.ltx2_abi_version.restype = i32
        l.ltx2_abi_version.argtypes = []
        have = l.ltx2_abi_version()
        if have != ABI_VERSION:
            # signatures changed between versions (arguments inserted mid-list): calling through mismatched ones would shift
            # pointers silently, so a library of another version is refused outright -- also under LTX2HIP_LIB
            raise NativeLibraryMissing(f"{path} reports ABI version {have}, this binding needs {ABI_VERSION}: rebuild it "
                                       "(`make -C ltx-2-mlx_amd/csrc`)")
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get(env) and not hasattr(l, name):
                continue                # an A/B build of the SAME ABI version that predates a newly ADDED entry: it just cannot be called
            try:
                fn = getattr(l, name)
            except AttributeError:      # the ABI and the header drifted apart without a version bump: still the 'rebuild it' message
                raise NativeLibraryMissing(f"{path} (ABI version {have}) does not export {name}: rebuild it (`make -C ltx-2-mlx_amd/csrc`)") from None
            fn.restype = res
            fn.argtypes = args
        _libs[key] = l
    return _libs[key]


def last_error() -> str:
    # every loaded build keeps its own thread-local message; the one that just failed is the non-empty one
    # (messages are cleared once read, so a handled failure of one build cannot resurface beside a later failure of another)
    msgs = [l.ltx2_last_error().decode("utf-8", "replace") for l in _libs.values()]
    for l in _libs.values():
        if hasattr(l, "ltx2_clear_error"):
            l.ltx2_clear_error()
    return " | ".join(m for m in msgs if m) or "unknown error"


def check(rc: int) -> None:
    """Map C-ABI status codes onto the exception types the reference raises (ValueError
    for bad arguments such as sigma == 0; RuntimeError otherwise)."""
    if rc == OK:
        return
    msg = last_error()
    if rc == E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(f"libltx2hip error {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw device pointer of a contiguous CUDA(ROCm) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libltx2hip operates on GPU tensors only (no CPU fallback)")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def exported_symbols() -> list:
    return list(SIGNATURES.keys())
