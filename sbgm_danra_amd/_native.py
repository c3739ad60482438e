"""ctypes binding of libsbgm_hip.so (C ABI: include/sbgm_hip.h).

There is deliberately no fallback: if the shared library is missing, or no ROCm device is present when a kernel
is requested, the caller gets an exception.  PyTorch is used only for device memory, streams and
torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsbgm_hip.so")

NONE, RELU, SILU, GELU = 0, 1, 2, 3
NORM_INSTANCE, NORM_GROUP = 0, 1
SAMPLER_EM, SAMPLER_PC = 0, 1

_vp, _i, _f, _i64, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64


class ModelConfig(C.Structure):
    _fields_ = [("struct_size", _i), ("n_lsm_channels", _i), ("n_topo_channels", _i), ("n_cond_channels", _i), ("time_embedding", _i),
                ("block_layers", _i * 4), ("n_heads", _i), ("num_classes", _i), ("last_fmap_channels", _i),
                ("decoder_norm", _i), ("gn_groups", _i), ("decoder_activation", _i), ("sigma", _f), ("decoder_transpose", _i)]


class SamplerArgs(C.Structure):
    _fields_ = [("kind", _i), ("B", _i), ("H", _i), ("W", _i), ("num_steps", _i), ("eps", _f), ("snr", _f),
                ("seed", _u64), ("use_graph", _i), ("bn_train", _i), ("y", _vp), ("cond_img", _vp), ("lsm_cond", _vp),
                ("topo_cond", _vp), ("noise", _vp), ("out", _vp), ("cfg_enabled", _i), ("cfg_scale", _f),
                ("cfg_scale_corrector", _f), ("tile_origins", _vp), ("domain_w", _i)]


class AssembleArgs(C.Structure):
    _fields_ = [("B", _i), ("HW", _i64), ("n_lr", _i), ("lr", _vp * 16), ("lr_channels", _i * 16), ("lsm", _vp),
                ("lsm_channels", _i), ("topo", _vp), ("topo_channels", _i), ("y", _vp), ("dropped", _vp), ("lr_out", _vp),
                ("lsm_out", _vp), ("topo_out", _vp), ("y_out", _vp)]


class PackDesc(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("Cout", _i), ("Cin", _i), ("KH", _i), ("KW", _i), ("cs", _i), ("nsteps", _i),
                ("transposed", _i), ("block_begin", _i)]


class AdamDesc(C.Structure):
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("numel", C.c_int64), ("block_begin", _i), ("reserved", _i)]


class Profile(C.Structure):
    _fields_ = [("ms_total_with_events", _f), ("ms_conv", _f), ("ms_conv_max", _f), ("flops_conv", C.c_double),
                ("flops_conv_max", C.c_double), ("n_conv", _i)]


class ConvArgs(C.Structure):
    _fields_ = [("x", _vp), ("w_packed", _vp), ("out", _vp), ("scale", _vp), ("bias", _vp), ("tbias", _vp),
                ("residual", _vp), ("B", _i), ("H", _i), ("W", _i), ("c_pad", _i), ("Cout", _i), ("KH", _i), ("KW", _i),
                ("stride", _i), ("pad", _i), ("act", _i), ("tbias_after_act", _i), ("tile_co", _i), ("tile_px", _i),
                ("splits", _i), ("waves_per_tile", _i), ("winograd", _i), ("in_dil", _i), ("out_h", _i), ("out_w", _i), ("ws", _vp),
                ("ws_floats", _i64), ("in_mode", _i), ("in_affine", _vp), ("in_skip", _vp), ("in_act", _i), ("w_wino", _vp), ("w_wino2d", _vp)]


# name -> (restype, argtypes); every symbol include/sbgm_hip.h declares
SIGNATURES = {
    "sbgm_last_error": (C.c_char_p, []),
    "sbgm_abi_version": (_i, []),
    "sbgm_model_create": (_i, [C.POINTER(ModelConfig), C.POINTER(_vp)]),
    "sbgm_model_config_size": (_i, []),
    "sbgm_model_destroy": (None, [_vp]),
    "sbgm_model_num_params": (_i, [_vp]),
    "sbgm_model_param_name": (C.c_char_p, [_vp, _i]),
    "sbgm_model_param_numel": (_i64, [_vp, _i]),
    "sbgm_model_set_param": (_i, [_vp, C.c_char_p, _vp, _i64, _vp]),
    "sbgm_model_get_param": (_i, [_vp, C.c_char_p, _vp, _i64, _vp]),
    "sbgm_model_check_complete": (_i, [_vp]),
    "sbgm_model_workspace_bytes": (_i64, [_vp]),
    "sbgm_model_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), _i, _i, _i, _i, _vp]),
    "sbgm_sampler_run": (_i, [_vp, C.POINTER(SamplerArgs), _vp]),
    "sbgm_pointwise_chain": (_i, [_vp, _vp, _i64, _i, C.POINTER(C.c_int), C.POINTER(C.c_float), _vp]),
    "sbgm_sample_extremes": (_i, [_vp, _i, _i64, _f, _vp, _vp, _vp]),
    "sbgm_assemble_conditions": (_i, [C.POINTER(AssembleArgs), _vp]),
    "sbgm_depth_to_space2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_space_to_depth2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_depth_to_space": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sbgm_space_to_depth": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sbgm_tconv_weight_to_oihw": (_i, [_vp, _vp, _i, _i, _vp]),
    "sbgm_extract_tiles": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sbgm_stitch_tiles": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sbgm_model_autotune": (_i, [_vp, _i, _i, _i, _vp]),
    "sbgm_model_tune_save": (_i, [_vp, C.c_char_p]),
    "sbgm_model_tune_load": (_i, [_vp, C.c_char_p]),
    "sbgm_model_profile_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(Profile), C.c_char_p, _vp]),
    "sbgm_event_create": (_i, [C.POINTER(_vp)]),
    "sbgm_event_record": (_i, [_vp, _vp]),
    "sbgm_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "sbgm_event_destroy": (_i, [_vp]),
    "sbgm_pack_input": (_i, [C.POINTER(_vp), C.POINTER(_i), _i, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_conv_packed_numel": (_i64, [_i, _i, _i, _i]),
    "sbgm_conv_pack_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sbgm_conv_wino2d_packed_numel": (_i64, [_i, _i]),
    "sbgm_conv_wino2d_pack_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sbgm_conv2d_fwd": (_i, [C.POINTER(ConvArgs), _vp]),
    "sbgm_conv2d_tune": (_i, [C.POINTER(ConvArgs), C.POINTER(C.c_int), _vp]),
    "sbgm_conv_pack_weights_batched": (_i, [_vp, _i, _i, _vp]),
    "sbgm_conv_pack_weights_batched_blocks": (_i, [_i, _i, _i, _i]),
    "sbgm_adam_step_blocks": (_i, [C.c_int64]),
    "sbgm_adam_step_batched": (_i, [_vp, _i, _i, _vp, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "sbgm_set_scratch_prezeroed": (_i, [_i]),
    "sbgm_conv_wino_packed_numel": (_i64, [_i, _i]),
    "sbgm_conv_wino_pack_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sbgm_upsample2x_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_groupnorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "sbgm_groupnorm_stats": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(_i), _vp]),
    "sbgm_groupnorm_finalize": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "sbgm_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sbgm_batchnorm_train_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "sbgm_batchnorm_train_stats": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "sbgm_batchnorm_train_apply": (_i, [_vp] * 8 + [_i, _i, _i, _i, _f, _f, _vp, C.c_double, _vp, _vp]),
    "sbgm_batchnorm_bwd_reduce": (_i, [_vp] * 5 + [_i, _vp, _i, _i, _i, _vp]),
    "sbgm_batchnorm_bwd_apply": (_i, [_vp] * 6 + [_i] + [_vp] * 6 + [C.c_double, _i, _i, _i, _vp]),
    "sbgm_mha_core_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_wgrad_defer": (_i, [_i]),
    "sbgm_wgrad_flush": (_i, [_vp]),
    "sbgm_wgrad_flush_pending": (_i, []),
    "sbgm_wgrad_discard": (_i, []),
    "sbgm_attn_qkv_fwd": (_i, [_vp] * 6 + [_i, _i, _f, _vp]),
    "sbgm_mha_core_dropout_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _u64, _u64, _vp]),
    "sbgm_mha_core_dropout_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _u64, _vp]),
    "sbgm_mha_dropout_mask": (_i, [_vp, _i, _i, _i, _f, _u64, _u64, _vp]),
    "sbgm_attn_tail_fwd": (_i, [_vp] * 11 + [_i, _i, _f, _vp]),
    "sbgm_time_proj_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sbgm_cout1_pack_weight": (_i, [_vp, _vp, _i, _vp]),
    "sbgm_conv3x3_cout1_fwd": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_act_inplace": (_i, [_vp, _i64, _i, _vp]),
    "sbgm_conv_pack_weight_dgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_conv8x8s2_dgrad_phase_weight": (_i, [_vp, _vp, _i, _i, _vp]),
    "sbgm_conv2d_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sbgm_conv2d_wgrad_bias": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sbgm_colsum": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sbgm_samplesum": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sbgm_groupnorm_bwd": (_i, [_vp] * 14 + [_i, _i, _i, _i, _vp]),
    "sbgm_batchnorm_bwd": (_i, [_vp] * 6 + [_i] + [_vp] * 5 + [_i, _i, _i, _vp]),
    "sbgm_layernorm_bwd": (_i, [_vp] * 6 + [_i, _i, _f, _vp, _vp]),
    "sbgm_fill_zero": (_i, [_vp, _i64, _vp]),
    "sbgm_mha_core_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_upsample2x_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_upsample_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sbgm_upsample_bilinear_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sbgm_conv3x3_cout1_bwd": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sbgm_time_proj_bwd": (_i, [_vp] * 7 + [_i, _i, _i, _vp]),
    "sbgm_time_proj_multi_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sbgm_time_proj_multi_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "sbgm_label_emb_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sbgm_act_fwd": (_i, [_vp, _vp, _i64, _i, _vp]),
    "sbgm_act_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "sbgm_dsm_loss_blocks": (_i, [_i64]),
    "sbgm_dsm_perturb": (_i, [_vp, _vp, _vp, _vp, _u64, _f, _f, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "sbgm_dsm_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "sbgm_dsm_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "sbgm_em_step": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _u64, _u64, _i64, _vp]),
    "sbgm_langevin_step": (_i, [_vp, _vp, _vp, _f, _vp, _u64, _u64, _i, _i64, _vp]),
    "sbgm_cfg_combine": (_i, [_vp, _vp, _vp, _f, _i64, _vp]),
    "sbgm_randn_scaled": (_i, [_vp, _f, _u64, _u64, _i64, _vp]),
}

ABI_VERSION = 4          # include/sbgm_hip.h: sbgm_abi_version()

_lib = None


class NativeError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libsbgm_hip.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C sbgm_danra_amd/csrc`). There is no CPU fallback for the HIP path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if l.sbgm_abi_version() != ABI_VERSION or l.sbgm_model_config_size() != C.sizeof(ModelConfig):
            raise NativeError(f"{LIB_PATH} has ABI version {l.sbgm_abi_version()} / a {l.sbgm_model_config_size()}-byte model config; this "
                              f"binding expects version {ABI_VERSION} / {C.sizeof(ModelConfig)} bytes: rebuild the library")
        _lib = l
    return _lib


# Weights can change without torch noticing: the native Adam step writes parameters through raw pointers and a replayed
# hipGraph re-runs captured BatchNorm running-statistics updates, neither of which bumps a tensor's version counter.  Every such
# writer calls bump_generation(); ScoreNet's engine compares generation() (with the version counters) before reusing its
# uploaded copy of the weights.
_generation = 0


def bump_generation() -> None:
    global _generation
    _generation += 1


def generation() -> int:
    return _generation


def check(rc: int) -> None:
    if rc != 0:
        raise NativeError(lib().sbgm_last_error().decode() or f"libsbgm_hip call failed with status {rc}")


def require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError("the SBGM HIP path needs tensors on a ROCm device (got a CPU tensor); "
                              "there is no CPU fallback — use the reference implementation for CPU runs")


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def f32c(t, name="tensor"):
    """contiguous fp32 view (copy only if needed)"""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
