"""ctypes binding of libddmi_hip.so (include/ddmi_hip.h).  The product path has NO fallback:
if the HIP library is missing or fails to load, importing a model class raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libddmi_hip.so")

PARITY_FP32, PERF_BF16, PARITY_BF16X3 = 0, 1, 2
LAYOUT_NCHW_F32, LAYOUT_NHWC_NATIVE = 0, 1
STEP_ANCESTRAL, STEP_DDIM, STEP_DDIM_LAST = 0, 1, 2


def numerics_code(name) -> int:
    if name in (PARITY_FP32, "fp32", "parity", "PARITY_FP32"):
        return PARITY_FP32
    if name in (PERF_BF16, "bf16", "perf", "PERF_BF16"):
        return PERF_BF16
    if name in (PARITY_BF16X3, "bf16x3", "PARITY_BF16X3"):
        return PARITY_BF16X3
    raise ValueError(f"numerics must be 'fp32' (parity), 'bf16' (perf) or 'bf16x3' (parity, convolutions as three bf16 products), got {name!r}")


def is_fp32_storage(numerics_code_value: int) -> bool:
    """Activations / planes are fp32 in both parity modes, bf16 in the perf mode."""
    return numerics_code_value != PERF_BF16


class UnetCfg(C.Structure):
    _fields_ = [("image_size", C.c_int), ("in_channels", C.c_int), ("model_channels", C.c_int),
                ("out_channels", C.c_int), ("num_res_blocks", C.c_int), ("n_mult", C.c_int),
                ("channel_mult", C.c_int * 8), ("n_attn", C.c_int), ("attention_resolutions", C.c_int * 8),
                ("num_head_channels", C.c_int), ("max_batch", C.c_int), ("numerics", C.c_int)]


class Decoder2dCfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("z_channels", C.c_int), ("resolution", C.c_int), ("out_ch", C.c_int),
                ("ch", C.c_int), ("num_res_blocks", C.c_int), ("n_mult", C.c_int), ("ch_mult", C.c_int * 8),
                ("n_attn", C.c_int), ("attn_resolutions", C.c_int * 8), ("n_hdbf", C.c_int),
                ("hdbf_resolutions", C.c_int * 8), ("latent_size", C.c_int), ("max_batch", C.c_int),
                ("numerics", C.c_int)]


class DecoderVCfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("z_channels", C.c_int), ("resolution", C.c_int), ("frames", C.c_int),
                ("ch", C.c_int), ("num_res_blocks", C.c_int), ("n_mult", C.c_int), ("ch_mult", C.c_int * 8),
                ("n_attn", C.c_int), ("attn_resolutions", C.c_int * 8), ("n_hdbf", C.c_int),
                ("hdbf_resolutions", C.c_int * 8), ("n_inter", C.c_int), ("inter_attn_resolutions", C.c_int * 8),
                ("attn_heads", C.c_int), ("max_batch", C.c_int), ("numerics", C.c_int)]


class McubesGrid(C.Structure):
    _fields_ = [("volume", C.c_void_p), ("dtype", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("pad", C.c_int), ("pad_value", C.c_float), ("isovalue", C.c_double)]


class Mlp2dCfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("out_ch", C.c_int), ("ch", C.c_int), ("latent_dim", C.c_int),
                ("numerics", C.c_int)]


class StepCoefs(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("sqrt_one_minus_ac", "sqrt_recip_ac", "sqrt_recipm1_ac", "coef1", "coef2",
                                         "noise_scale", "sqrt_ac_next", "c_dir", "sigma")]


# every symbol include/ddmi_hip.h declares: name -> (restype, argtypes)
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
_I64P = C.POINTER(C.c_int64)
SYMBOLS = {
    "ddmi_init": (_I, [_I]),
    "ddmi_last_error": (C.c_char_p, []),
    "ddmi_version": (C.c_char_p, []),
    "ddmi_profile_enable": (_I, [_I]),
    "ddmi_profile_collect": (_I, [C.c_char_p, C.c_uint64]),
    "ddmi_unet_create": (_I, [C.POINTER(UnetCfg), C.POINTER(_VP)]),
    "ddmi_unet_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_unet_finalize": (_I, [_VP]),
    "ddmi_unet_forward": (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    "ddmi_unet_set_uniform_t": (_I, [_VP, _I]),
    "ddmi_unet_destroy": (_I, [_VP]),
    "ddmi_unet_memory": (_I, [_VP, C.POINTER(C.c_uint64 * 3)]),
    "ddmi_ddpm_step": (_I, [_VP, _VP, _VP, _VP, C.POINTER(StepCoefs), _I, _I, _VP, _VP, _I, _I, _I, _VP]),
    "ddmi_ddpm_step_rng": (_I, [_VP, _VP, _VP, C.c_uint32, _VP, C.POINTER(StepCoefs), _I, _I, _VP, _VP, _I, _I, _I, _VP]),
    "ddmi_randn_fill": (_I, [_VP, _VP, _I, C.c_int64, C.c_uint32, _VP]),
    "ddmi_decoder2d_create": (_I, [C.POINTER(Decoder2dCfg), C.POINTER(_VP)]),
    "ddmi_decoder2d_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_decoder2d_finalize": (_I, [_VP]),
    "ddmi_decoder2d_num_planes": (_I, [_VP, C.POINTER(_I), C.POINTER(_I * 8)]),
    "ddmi_decoder2d_forward": (_I, [_VP, _VP, C.POINTER(_VP), _I, _I, _VP]),
    "ddmi_decoder2d_destroy": (_I, [_VP]),
    "ddmi_mlp2d_create": (_I, [C.POINTER(Mlp2dCfg), C.POINTER(_VP)]),
    "ddmi_mlp2d_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_mlp2d_finalize": (_I, [_VP]),
    "ddmi_hdpe_mlp2d": (_I, [_VP, C.POINTER(_VP), C.POINTER(_I), _I, _VP, _F, _VP, _VP, _I, _I, _VP]),
    "ddmi_hdpe_mlp2d_rng": (_I, [_VP, C.POINTER(_VP), C.POINTER(_I), _I, _VP, _F, _VP, _VP, _I, _I, _VP]),
    "ddmi_mlp2d_noise_nonzero": (_I, [_VP, C.POINTER(_I)]),
    "ddmi_mlp2d_destroy": (_I, [_VP]),
    "ddmi_decoder3d_create": (_I, [C.POINTER(Decoder2dCfg), C.POINTER(_I), _I, C.POINTER(_VP)]),
    "ddmi_decoder3d_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_decoder3d_finalize": (_I, [_VP]),
    "ddmi_decoder3d_num_scales": (_I, [_VP, C.POINTER(_I), C.POINTER(_I * 4)]),
    "ddmi_decoder3d_forward": (_I, [_VP, _VP, C.POINTER(_VP), _I, _I, _VP]),
    "ddmi_decoder3d_destroy": (_I, [_VP]),
    "ddmi_mlp3d_create": (_I, [C.POINTER(Mlp2dCfg), C.POINTER(_VP)]),
    "ddmi_mlp3d_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_mlp3d_finalize": (_I, [_VP]),
    "ddmi_hdpe_mlp3d": (_I, [_VP, C.POINTER(_VP), C.POINTER(_I), _I, _VP, _VP, _I, _I, _VP]),
    "ddmi_mlp3d_destroy": (_I, [_VP]),
    "ddmi_unet_set_graph": (_I, [_VP, _I]),
    "ddmi_unet3_create": (_I, [C.POINTER(UnetCfg), _I, _I, _I, C.POINTER(_VP)]),
    "ddmi_decoderv_create": (_I, [C.POINTER(DecoderVCfg), C.POINTER(_VP)]),
    "ddmi_decoderv_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_decoderv_finalize": (_I, [_VP]),
    "ddmi_decoderv_num_scales": (_I, [_VP, C.POINTER(_I), C.POINTER(_I * 4), C.POINTER(_I)]),
    "ddmi_decoderv_forward": (_I, [_VP, _VP, C.POINTER(_VP), _I, _I, _VP]),
    "ddmi_decoderv_destroy": (_I, [_VP]),
    "ddmi_mlpv_create": (_I, [C.POINTER(Mlp2dCfg), C.POINTER(_VP)]),
    "ddmi_mlpv_load_weight": (_I, [_VP, C.c_char_p, _VP, _I64P, _I]),
    "ddmi_mlpv_finalize": (_I, [_VP]),
    "ddmi_hdpe_mlpv": (_I, [_VP, C.POINTER(_VP), C.POINTER(_I), _I, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_mlpv_destroy": (_I, [_VP]),
    "ddmi_op_conv2d": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "ddmi_op_conv2d_skip": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "ddmi_bench_conv": (_I, [_I] * 13 + [C.POINTER(C.c_double), C.POINTER(_I)]),
    "ddmi_debug_conv_config": (_I, [_I]),
    "ddmi_debug_halo_half": (_I, [_I]),
    "ddmi_op_attention_block": (_I, [_VP] * 8 + [_I] * 6 + [C.POINTER(_I)]),
    "ddmi_debug_attn_fused": (_I, [_I]),
    "ddmi_debug_gn_fuse": (_I, [_I]),
    "ddmi_debug_small_batch": (_I, [_I]),
    "ddmi_debug_small_gn_launches": (_I, [C.POINTER(_I)]),
    "ddmi_op_groupnorm_small": (_I, [_VP] * 5 + [_I] * 4 + [_F, _I]),
    "ddmi_debug_conv_gn_fusable": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "ddmi_op_conv1x1_cat": (_I, [_VP] * 5 + [_I] * 7 + [C.POINTER(_I)]),
    "ddmi_op_conv_gn": (_I, [_VP] * 7 + [_I] * 5 + [C.c_float] + [_I] * 3 + [C.POINTER(_I)]),
    "ddmi_unet_async_status": (_I, [_VP, C.POINTER(_I)]),
    "ddmi_unet_async_status_on": (_I, [_VP, _VP, C.POINTER(_I)]),
    "ddmi_unet_debug_inject_gn_timeout": (_I, [_VP]),
    "ddmi_bench_peak_mfma": (_I, [C.c_double, C.POINTER(C.c_double)]),
    "ddmi_bench_peak_copy": (_I, [C.c_double, C.POINTER(C.c_double)]),
    "ddmi_bench_attention_block": (_I, [_I] * 5 + [C.POINTER(C.c_double), C.POINTER(_I)]),
    "ddmi_op_attention_legacy_hd": (_I, [_VP, _VP, _I, _I, _I, _I, _I]),
    "ddmi_op_attention_mh": (_I, [_VP, _VP, _VP, _VP] + [_I] * 7),
    "ddmi_op_attention_mh_prescaled": (_I, [_VP, _VP, _VP, _VP] + [_I] * 6),
    "ddmi_op_groupnorm": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _F, _I, _I]),
    "ddmi_op_attention_legacy": (_I, [_VP, _VP, _I, _I, _I, _I]),
    "ddmi_op_timestep_embedding": (_I, [_VP, _VP, _I, _I]),
    "ddmi_mcubes_workspace_bytes": (_I, [C.POINTER(McubesGrid), C.POINTER(C.c_size_t)]),
    "ddmi_mcubes_count": (_I, [C.POINTER(McubesGrid), _VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _VP]),
    "ddmi_mcubes_emit": (_I, [C.POINTER(McubesGrid), _VP, _VP, _VP, _I, C.c_double, _VP]),
    "ddmi_image_to_u8": (_I, [_VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_fused_bias_act": (_I, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int64, _I, _I, _I, C.c_float, C.c_float, _I, _VP]),
    # training (include/ddmi_hip.h "Training")
    "ddmi_train_create": (_I, [_I, C.POINTER(_VP)]),
    "ddmi_train_destroy": (_I, [_VP]),
    "ddmi_train_to_nhwc": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "ddmi_train_to_nchw": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "ddmi_train_conv_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _VP, _VP] + [_I] * 8 + [_VP]),
    "ddmi_train_conv_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP] + [_I] * 8 + [_VP]),
    "ddmi_train_gn_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _F, _I, _VP]),
    "ddmi_train_gn_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_gn_bwd_res": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_cat2": (_I, [_VP, _VP, _VP, _VP, C.c_int64, _I, _I, _VP]),
    "ddmi_train_split2": (_I, [_VP, _VP, _VP, _VP, C.c_int64, _I, _I, _VP]),
    "ddmi_train_attn_fwd": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_attn_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_attn_lse_ok": (_I, [_VP, _I, _I]),
    "ddmi_train_attn_fwd_lse": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_attn_bwd_lse": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_attn_fwd_lse_scaled": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _F, _VP]),
    "ddmi_train_attn_bwd_lse_scaled": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _F, _VP]),
    "ddmi_train_linear_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_linear_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "ddmi_train_accumulate_next": (_I, [_VP, _I]),
    "ddmi_train_arena_generation": (_I, [_VP, _VP]),
    "ddmi_train_matmul": (_I, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int64, C.c_int64, _VP, _VP, _I, _I, _I, _VP]),
    "ddmi_train_timestep_embedding": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "ddmi_train_q_sample": (_I, [_VP, _VP, _VP, _VP, _VP, _I, C.c_int64, _VP]),
    "ddmi_train_eps_loss": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "ddmi_train_adamw": (_I, [_VP, _VP, _VP, _VP, C.c_int64, _F, _F, _F, _F, _F, _I, _F, _VP]),
    "ddmi_train_ema": (_I, [_VP, _VP, C.c_int64, _F, _VP]),
}

_lib: Optional[C.CDLL] = None
_inited_devices = set()


class DdmiError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """The loaded library; raises (loudly) when it has not been built - there is no CPU/torch fallback."""
    global _lib
    if _lib is None:
        path = os.environ.get("DDMI_LIB_PATH", LIB_PATH)  # diagnostics: A/B two builds of the library on one GPU box
        if path != LIB_PATH:
            globals()["LIB_PATH"] = path
        if not os.path.exists(LIB_PATH):
            raise DdmiError(f"{LIB_PATH} is missing: build it with `python -m ddmi_amd.build` "
                            "(hipcc --offload-arch=gfx950); ddmi_amd has no fallback path")
        default_plans = os.path.join(HERE, "tuned", "conv_plans_mi355x.txt")
        if os.path.exists(default_plans):   # read-only plans shipped with the package (net.hip: DDMI_TUNE_DEFAULT)
            os.environ.setdefault("DDMI_TUNE_DEFAULT", default_plans)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().ddmi_last_error()
        raise DdmiError(f"ddmi_hip error {rc}: {msg.decode() if msg else '?'}")


def init_device(index: int) -> None:
    if index not in _inited_devices:
        check(lib().ddmi_init(index))
        _inited_devices.add(index)


def profile_enable(every_n_forwards: int) -> None:
    check(lib().ddmi_profile_enable(int(every_n_forwards)))


def profile_collect() -> dict:
    import json
    buf = C.create_string_buffer(1 << 16)
    check(lib().ddmi_profile_collect(buf, len(buf)))
    return json.loads(buf.value.decode())


def measured_peaks(mfma_seconds: float = 1.0, copy_gib: float = 2.0) -> dict:
    """The two on-box microbenchmarks (k_microbench.hip): the dense bf16 MFMA rate and the device-copy bandwidth THIS GPU sustains."""
    tf, tb = C.c_double(0.0), C.c_double(0.0)
    check(lib().ddmi_bench_peak_mfma(float(mfma_seconds), C.byref(tf)))
    check(lib().ddmi_bench_peak_copy(float(copy_gib), C.byref(tb)))
    return {"mfma_bf16_tflops": round(tf.value, 1), "hbm_copy_tbs": round(tb.value, 3),
            "how": f"pure v_mfma_f32_16x16x32_bf16 loop on random register operands, every CU, 2 waves/SIMD, ~{mfma_seconds:g} s; "
                   f"16-byte-per-lane copy of {copy_gib:g} GiB x5, (read + written) bytes / s"}


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def load_weights(load_fn, handle, state, prefix_strip: str = "", only_prefixes=None) -> int:
    """Hands every fp32 tensor of a state-dict-like mapping to `load_fn(handle, key, ptr, shape, ndim)`."""
    import numpy as np
    n = 0
    for key, val in state.items():
        if only_prefixes is not None and not any(key.startswith(p) for p in only_prefixes):
            continue
        k = key[len(prefix_strip):] if prefix_strip and key.startswith(prefix_strip) else key
        if hasattr(val, "detach"):
            val = val.detach().float().cpu().numpy()
        arr = np.ascontiguousarray(np.asarray(val, dtype=np.float32))
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        check(load_fn(handle, k.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
        n += 1
    return n
