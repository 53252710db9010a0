"""ctypes binding of libpnpi.so (include/pnpi.h).  There is no CPU fallback: if the HIP library is missing or a call
fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpnpi.so")

PNPI_OK, PNPI_EINVAL, PNPI_ESHAPE, PNPI_EHIP, PNPI_ESTATE, PNPI_ENOMEM = 0, -1, -2, -3, -4, -5


class ModelConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("n_blocks", C.c_int),
        ("block_out_channels", C.c_int * 4), ("block_has_attn", C.c_int * 4),
        ("layers_per_block", C.c_int), ("heads", C.c_int), ("cross_dim", C.c_int), ("ctx_len", C.c_int),
        ("sample_size", C.c_int), ("norm_groups", C.c_int), ("n_train_timesteps", C.c_int),
        ("vae_in_channels", C.c_int), ("vae_latent_channels", C.c_int), ("vae_n_blocks", C.c_int),
        ("vae_block_out_channels", C.c_int * 4), ("vae_layers_per_block", C.c_int), ("vae_norm_groups", C.c_int),
        ("clip_layers", C.c_int), ("clip_heads", C.c_int), ("clip_intermediate", C.c_int), ("clip_vocab", C.c_int),
    ]


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class CtrlDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("n_alpha_rows", C.c_int),
        ("cross_alpha_host", C.POINTER(C.c_float)), ("mapper_host", C.POINTER(C.c_float)),
        ("alphas_host", C.POINTER(C.c_float)), ("equalizer_host", C.POINTER(C.c_float)),
        ("self_replace_lo", C.c_int), ("self_replace_hi", C.c_int), ("self_replace_max_tokens", C.c_int),
        ("lb_enabled", C.c_int), ("lb_start", C.c_int), ("lb_threshold", C.c_float),
        ("lb_alpha_host", C.POINTER(C.c_float)),
        ("masa_start_step", C.c_int), ("masa_start_layer", C.c_int),
        ("lb_sub_alpha_host", C.POINTER(C.c_float)), ("lb_threshold_sub", C.c_float),
        ("masa_layer_mask", C.c_uint), ("masa_n_steps", C.c_int), ("masa_step_on_host", C.POINTER(C.c_ubyte)),
    ]


class ReconDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("ref_image", C.c_void_p), ("recon_lr", C.c_float), ("recon_t", C.c_int), ("dilate_mask", C.c_int),
                ("inv_x_stars", C.c_void_p)]

    @classmethod
    def make(cls, ref_image, recon_lr, recon_t, dilate_mask=0, inv_x_stars=None):
        return cls(C.sizeof(cls), ref_image, float(recon_lr), int(recon_t), int(dilate_mask), inv_x_stars)


ATTN_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)


class Counters(C.Structure):
    _fields_ = [("unet_sample_forwards", C.c_uint64), ("unet_calls", C.c_uint64), ("vae_encodes", C.c_uint64),
                ("vae_decodes", C.c_uint64), ("executed_gemm_flops", C.c_double), ("executed_attn_flops", C.c_double),
                ("text_kv_rows", C.c_uint64), ("unet_sample_forwards_cached_kv", C.c_uint64), ("unet_backward_rows", C.c_uint64)]


KC_NAMES = ["igemm128", "igemm64", "igemm64_splitk", "attn_flash", "attn_cross_edit", "groupnorm", "layernorm", "geglu", "softmax", "igemm_wide"]


class KernelStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/pnpi.h declares: name -> (restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
SYMBOLS = {
    "pnpi_config_sd1": (None, [C.POINTER(ModelConfig)]),
    "pnpi_create": (_i, [C.POINTER(_vp), C.POINTER(ModelConfig), _i, _vp, _i, _i]),
    "pnpi_create_shared": (_i, [C.POINTER(_vp), _vp, _vp, _i, _i]),
    "pnpi_destroy": (None, [_vp]),
    "pnpi_last_error": (C.c_char_p, [_vp]),
    "pnpi_load_weights": (_i, [_vp, C.POINTER(NamedTensor), _i]),
    "pnpi_missing_weights": (_i, [_vp, C.c_char_p, _sz]),
    "pnpi_weight_arena": (_i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "pnpi_mark_all_loaded": (_i, [_vp]),
    "pnpi_set_scheduler": (_i, [_vp, C.POINTER(C.c_float), _i, _f]),
    "pnpi_get_counters": (_i, [_vp, C.POINTER(Counters)]),
    "pnpi_reset_counters": (_i, [_vp]),
    "pnpi_clock_probe": (_i, [_vp, _i, _fp, _fp]),
    "pnpi_profile_begin": (_i, [_vp]),
    "pnpi_profile_end": (_i, [_vp, C.POINTER(KernelStats)]),
    "pnpi_unet_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, C.POINTER(CtrlDesc), _i, _vp]),
    "pnpi_text_kv_precompute": (_i, [_vp, _vp, _i]),
    "pnpi_set_attention_callback": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "pnpi_local_blend": (_i, [_vp, _vp, _i, _i]),
    "pnpi_vae_encode": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pnpi_vae_decode": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pnpi_image2latent": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pnpi_latent2image": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pnpi_ddim_next_step": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "pnpi_ddim_prev_step": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "pnpi_ddim_prev_step_recon": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp, _f, _vp, _vp, _vp]),
    "pnpi_cfg_ddim_prev": (_i, [_vp, _vp, _vp, _i, _i, _sz, _f, _i, _i, _vp, _i, _vp, _f, _vp, _vp, _vp, _i, C.POINTER(ReconDesc)]),
    "pnpi_prox_threshold": (_i, [_vp, _vp, _i, _i, _sz, _f, _vp]),
    "pnpi_text_encode": (_i, [_vp, _vp, _i, _vp]),
    "pnpi_ddim_invert": (_i, [_vp, _vp, _i, _vp, _i, _ip, _vp]),
    "pnpi_ddim_invert_cfg": (_i, [_vp, _vp, _i, _vp, _vp, _f, _i, _ip, _vp]),
    "pnpi_offset_calculate": (_i, [_vp, _vp, _i, _vp, _i, _ip, _f, _fp, _vp]),
    "pnpi_edit_loop": (_i, [_vp, _vp, _i, _vp, _vp, _i, C.POINTER(CtrlDesc), _i, _ip, _f, _i, _f, C.POINTER(ReconDesc), _vp]),
    "pnpi_direct_edit": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(CtrlDesc), _i, _i, _ip, _f, _fp, _vp, _vp]),
    "pnpi_direct_edit_pruned": (_i, [_vp, _vp, _i, _vp, C.POINTER(CtrlDesc), _i, _ip, _f, _vp]),
    "pnpi_op_conv": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i]),
    "pnpi_op_conv_stats": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _ip]),
    "pnpi_set_tuning": (_i, [C.c_char_p, _i]),
    "pnpi_tile_table_lookup": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pnpi_op_gemm": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i]),
    "pnpi_op_gemm_geglu": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i]),
    "pnpi_op_groupnorm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "pnpi_op_layernorm": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "pnpi_op_geglu": (_i, [_vp, _vp, _i, _i, _vp]),
    "pnpi_op_softmax_rows": (_i, [_vp, _vp, _i, _i, _i]),
    "pnpi_op_layernorm_bwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "pnpi_op_groupnorm_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp]),
    "pnpi_op_geglu_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "pnpi_op_softmax_bwd_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pnpi_op_accumulate": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "pnpi_op_sumpool2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pnpi_op_zero_stuff2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pnpi_op_repack_dgrad": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pnpi_op_null_text_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp]),
    "pnpi_op_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f]),
    "pnpi_op_attention_bwd": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, C.c_size_t]),
    "pnpi_op_attention_bwd_scratch_bytes": (C.c_size_t, [_i, _i, _i]),
    "pnpi_edit_loop_uncond_steps": (_i, [_vp, _vp, _i, _vp, _vp, _i, C.POINTER(C.c_int), _f, _i, _f, _vp, _i, _vp]),
    "pnpi_edit_loop_uncond_steps_recon": (_i, [_vp, _vp, _i, _vp, _vp, _i, C.POINTER(C.c_int), _f, _i, _f, _vp, _i, C.POINTER(ReconDesc), _vp]),
    "pnpi_unet_context_grad": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "pnpi_null_text_optimize": (_i, [_vp, _vp, _vp, _vp, _i, C.POINTER(C.c_int), _f, _i, _f, _vp, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "pnpi_null_latent_calculate": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_int), _f, _i, _f, _vp, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "pnpi_op_attention": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i]),
    "pnpi_op_cross_edit": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp,
                                _vp, _vp, _vp, _i, _i]),
    "pnpi_op_local_blend": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _i]),
    "pnpi_op_local_blend_sub": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _i]),
}

_lib = None


class PnpiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libpnpi status %d: %s" % (status, message))
        self.status = status


def load_library(path=None):
    """dlopen libpnpi.so and bind every symbol of include/pnpi.h.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch ships its own libamdhip64; it must be the first (and only) HIP runtime mapped into the process, otherwise this
    # library would initialise a second runtime that sees no device.
    import torch  # noqa: F401
    p = path or os.environ.get("PNPI_LIBRARY") or LIB_PATH      # PNPI_LIBRARY: e.g. csrc/libpnpi_ablations.so for tools/pp_ablate.py
    if not os.path.exists(p):
        raise ImportError(
            "libpnpi.so not found at %s -- build it with `python -m pnpinversion_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the native path." % p)
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # PNPI_TUNE="key=value,key=value": process-wide kernel tuning knobs (pnpi_set_tuning) applied at load -- whole test files / the bench
    # under a non-default kernel variant without touching their code
    for kv in filter(None, os.environ.get("PNPI_TUNE", "").split(",")):
        k, v = kv.split("=")
        if lib.pnpi_set_tuning(k.strip().encode(), int(v)) != 0:
            raise ValueError("PNPI_TUNE: unknown tuning key %r, or a value this build does not carry (ablation selectors need "
                             "`python -m pnpinversion_amd.build --ablations` + PNPI_LIBRARY)" % k)
    if path is None:
        _lib = lib
    return lib


def check(lib, ctx, status):
    if status != 0:
        msg = lib.pnpi_last_error(ctx)
        raise PnpiError(status, msg.decode() if msg else "?")
