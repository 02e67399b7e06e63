"""ctypes binding of libsplitvae_hip.so (include/splitvae.h).

The HIP library IS the product path: if it is missing or fails to load, importing this module
raises -- there is no CPU/PyTorch fallback anywhere in split_vae_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("SV_LIB_NAME", "libsplitvae_hip.so"))   # override: kernel A/B builds

SV_F32, SV_BF16 = 0, 1
SV_ACT_NONE, SV_ACT_RELU, SV_ACT_ELU = 0, 1, 2
PHASE_PREP, PHASE_FWD_ENCODERS, PHASE_FWD_DECODERS, PHASE_LOSS = 1, 2, 4, 8
PHASE_BWD_DECODERS, PHASE_BWD_ENC_HEADS, PHASE_BWD_ENC_CONVS, PHASE_ADAM = 16, 32, 64, 128
PHASE_FORWARD, PHASE_BACKWARD, PHASE_ALL = 6, 112, 255
PHASE_INPUTS_STAGED = 512  # modifier: in8_x / in8_xh were written by sv_scramble_gather_staged (include/splitvae.h)
PHASE_NO_RECON = 256      # modifier: the fused-loss training step does not store out6_x / out6_xh (include/splitvae.h)
PHASE_BUCKET_EVENTS = 1024  # modifier: record the gradient-bucket events of the data-parallel step (include/splitvae.h)
PHASE_INFER = PHASE_PREP | PHASE_FORWARD

STATUS = {0: "SV_OK", -1: "SV_E_BADARG", -2: "SV_E_UNSUPPORTED", -3: "SV_E_WORKSPACE", -4: "SV_E_STATE"}
STATUS_BADARG, STATUS_UNSUPPORTED, STATUS_WORKSPACE, STATUS_STATE = -1, -2, -3, -4


class SplitVaeError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "act", "dtype", "ldx", "ldy", "y_f32", "ups_in")]


class LGVaeDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("global_latent", C.c_int32),
                ("local_latent", C.c_int32), ("dtype", C.c_int32), ("beta", C.c_float),
                ("external_global_encoder", C.c_int32)]


class StepArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("images6", C.c_void_p), ("eps_x", C.c_void_p), ("eps_x_hat", C.c_void_p),
                ("seed", C.c_uint64), ("step", C.c_uint64), ("sample_offset", C.c_int64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("t", C.c_int64), ("grad_scale", C.c_float), ("phases", C.c_int32),
                ("accumulate_metrics", C.c_int32)]


class TapeNode(C.Structure):
    """sv_tape_node (include/splitvae.h)."""
    _fields_ = ([(n, C.c_int32) for n in ("kind", "x", "y", "t2", "t3", "t4", "t5", "t6", "xo", "yo", "o2", "o3", "n", "rep", "op", "act")] +
                [("p0", C.c_float), ("p1", C.c_float), ("w_off", C.c_int64), ("b_off", C.c_int64)] +
                [(n, C.c_int32) for n in ("B", "H", "W", "C", "Cout", "k", "stride", "Ho", "Wo", "Hc", "Wc", "inverse", "training",
                                          "loss_idx", "dyn_idx", "mode", "R", "stream_id", "group", "lane")])


class TapeRunArgs(C.Structure):
    """sv_tape_run_args (include/splitvae.h)."""
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("loss_weights", C.POINTER(C.c_float)), ("n_weights", C.c_int32),
                ("dyn", C.c_float * 8), ("seed", C.c_uint64), ("step", C.c_uint64), ("pinned_noise", C.c_int32), ("phases", C.c_int32),
                ("accumulate_metrics", C.c_int32), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("n_params", C.c_int64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float), ("t", C.c_int64),
                ("clipnorm", C.c_float), ("tensor_off", C.c_void_p), ("n_tensors", C.c_int32), ("norm_ws", C.c_void_p)]


TAPE_DENSE, TAPE_CONV, TAPE_UNARY, TAPE_SAMPLE, TAPE_LOGITNOISE, TAPE_UPSAMPLE, TAPE_STN, TAPE_RENDER, TAPE_ZPRES, TAPE_LOSS, TAPE_NOISE = range(11)
TAPE_COPY, TAPE_RELU, TAPE_SIGMOID, TAPE_SOFTPLUS, TAPE_CLAMP, TAPE_SCALE = range(6)
TAPE_PHASE_FORWARD, TAPE_PHASE_BACKWARD, TAPE_PHASE_ADAM = 1, 2, 4
TAPE_MAX_LOSS = 16


class GmDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("latent", C.c_int32), ("y_size", C.c_int32),
                ("tau", C.c_float), ("dtype", C.c_int32)]


class GmArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("in8_x", C.c_void_p), ("zcat", C.c_void_p), ("ldz", C.c_int32),
                ("gz", C.c_void_p), ("ld_gz", C.c_int32), ("eps", C.c_void_p), ("u", C.c_void_p), ("keep1", C.c_void_p),
                ("keep5", C.c_void_p), ("training", C.c_int32), ("beta", C.c_float), ("alpha", C.c_float),
                ("seed", C.c_uint64), ("step", C.c_uint64), ("sample_offset", C.c_int64)]


# every symbol include/splitvae.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
SYMBOLS = {
    "sv_version": (C.c_char_p, []),
    "sv_set_deterministic": (C.c_int, [_i32]),
    "sv_get_deterministic": (C.c_int, []),
    "sv_scramble_gather": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "sv_scramble_gather_staged": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_random_perm": (C.c_int, [_vp, _i32, _i32, _u64, _u64, _i64, _vp]),
    "sv_dlogistic_nll": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _f, _i32, _i32, _i32, _vp, _vp]),
    "sv_dlogistic_nll_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "sv_reparam_kl_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32,
                                    _u64, _u64, _i32, _i64, _vp]),
    "sv_reparam_kl_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _f, _vp, _i32, _i32, _i32, _vp]),
    "sv_act_fwd": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _f, _vp, _vp, _u64, _u64, _i32, _i64,
                             _i32, _vp]),
    "sv_act_bwd": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _f, _vp, _vp, _i32, _i32, _i64,
                             _i32, _vp]),
    "sv_add": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp]),
    "sv_gm_metrics": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _f, _f, _vp, _vp]),
    "sv_gumbel_softmax_fwd": (C.c_int, [_vp, _i32, _vp, _vp, _f, _vp, _vp, _i32, _i32, _i32, _i32, _u64, _u64, _i64, _vp]),
    "sv_gumbel_softmax_bwd": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _f, _f, _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "sv_gm_head_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32,
                                 _i32, _u64, _u64, _i64, _vp]),
    "sv_gm_head_bwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "sv_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i64, _f, _vp]),
    "sv_upsample2x_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_upsample2x_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_stn_sample_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_spair_loss_dyn": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _vp, C.c_float, _vp]),
    "sv_spair_zpres_kl_dyn": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _vp, C.c_float, C.c_float, _vp]),
    "sv_adam_step_clipnorm_dyn": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                            _i64, _vp, C.c_float, _vp]),
    "sv_adam_step_clipnorm_ptrs": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                             _i64, _vp, C.c_float, _vp]),
    "sv_adam_alpha": (C.c_float, [C.c_float, C.c_float, C.c_float, _i64]),
    "sv_spair_loss": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, C.c_float, _vp]),
    "sv_stn_bwd_overwrites": (C.c_int, [_i32, _i32, _i32, _i32]),
    "sv_stn_sample_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_spair_render_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_spair_render_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_spair_render_bwd_workspace_floats": (_i64, [_i32, _i32, _i32]),
    "sv_spair_render_bwd_ws": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "sv_spair_zpres_kl": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _f, _f, _vp]),
    "sv_adam_step_clipnorm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _f, _f, _f, _f, _f, _i64, _f, _vp]),
    "sv_conv2d_wprep_elems": (_i64, [C.POINTER(ConvDesc), _i32]),
    "sv_conv2d_prep_weights": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "sv_conv2d_nhwc_fwd": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "sv_conv2d_fwd_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "sv_conv2d_nhwc_fwd_ws": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sv_conv2d_nhwc_dgrad": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i32, _vp]),
    "sv_conv2d_nhwc_dgrad_lowres": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "sv_conv2d_dgrad_lowres_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "sv_conv2d_nhwc_dgrad_lowres_ws": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sv_conv2d_nhwc_wgrad": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "sv_conv2d_wgrad_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "sv_conv2d_nhwc_wgrad_ws": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sv_conv2d_wgrad_poly_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "sv_conv2d_nhwc_wgrad_poly": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sv_crc32c": (C.c_uint32, [_vp, _i64]),
    "sv_masked_crc32c": (C.c_uint32, [_vp, _i64]),
    "sv_gm_param_count": (_i64, [C.POINTER(GmDesc)]),
    "sv_gm_param_info": (C.c_int, [C.POINTER(GmDesc), _i32, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64 * 4), C.c_char_p]),
    "sv_gm_encoder_create": (C.c_int, [C.POINTER(GmDesc), C.POINTER(_vp)]),
    "sv_gm_encoder_destroy": (None, [_vp]),
    "sv_gm_encoder_workspace_bytes": (_i64, [_vp]),
    "sv_gm_encoder_bind": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sv_gm_encoder_buffer": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "sv_gm_encoder_prep": (C.c_int, [_vp, _vp, _vp]),
    "sv_gm_encoder_forward": (C.c_int, [_vp, C.POINTER(GmArgs), _vp]),
    "sv_gm_encoder_backward": (C.c_int, [_vp, C.POINTER(GmArgs), _vp]),
    "sv_gm_encoder_y_kl": (C.c_int, [_vp, _vp]),
    "sv_dense_f32_fwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_dense_f32_dgrad": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sv_dense_f32_wgrad": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "sv_tape_create": (C.c_int, [C.POINTER(_vp), _i32, _i32]),
    "sv_tape_destroy": (None, [_vp]),
    "sv_tape_tensor": (_i32, [_vp, _i64, _i32, _i32, _i32]),
    "sv_tape_view": (_i32, [_vp, _i32, _i64, _i32, _i32]),
    "sv_tape_add": (C.c_int, [_vp, C.POINTER(TapeNode)]),
    "sv_tape_schedule": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp]),
    "sv_tape_set_report": (C.c_int, [_vp, C.POINTER(C.c_float), _i32]),
    "sv_tape_finalize": (C.c_int, [_vp]),
    "sv_tape_workspace_bytes": (_i64, [_vp]),
    "sv_tape_bind": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sv_tape_tensor_info": (C.c_int, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "sv_tape_loss_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "sv_tape_run": (C.c_int, [_vp, C.POINTER(TapeRunArgs), _vp]),
    "sv_comm_unique_id": (C.c_int, [_vp]),
    "sv_comm_init": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "sv_comm_allreduce": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sv_comm_allreduce_ranges": (C.c_int, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _i32, _vp]),
    "sv_comm_destroy": (C.c_int, [_vp]),
    "sv_lgvae_param_count": (_i64, [C.POINTER(LGVaeDesc)]),
    "sv_lgvae_param_info": (C.c_int, [C.POINTER(LGVaeDesc), _i32, C.POINTER(_i64), C.POINTER(_i32),
                                      C.POINTER(_i64 * 4), C.c_char_p]),
    "sv_lgvae_plan_create": (C.c_int, [C.POINTER(LGVaeDesc), C.POINTER(_vp)]),
    "sv_lgvae_plan_destroy": (None, [_vp]),
    "sv_lgvae_workspace_bytes": (_i64, [_vp]),
    "sv_lgvae_plan_bind": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sv_lgvae_buffer": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "sv_lgvae_step": (C.c_int, [_vp, C.POINTER(StepArgs), _vp]),
    "sv_lgvae_graph_enable": (C.c_int, [_vp, _i32]),
    "sv_lgvae_bucket_wait": (C.c_int, [_vp, _i32, _vp]),
    "sv_lgvae_plan_debug": (C.c_int, [_vp, C.c_char_p, _i64]),
    "sv_side_stream": (C.c_int, [_i32, _vp]),
    "sv_lgvae_graph_count": (C.c_int, [_vp]),
    "sv_lgvae_profile_enable": (C.c_int, [_vp, _i32]),
    "sv_lgvae_profile_filter": (C.c_int, [_vp, C.c_char_p]),
    "sv_lgvae_profile_read": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sv_lgvae_profile_read_issued": (C.c_int, [_vp, _i32, _vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises SplitVaeError when it is absent: build it with
    `python -m split_vae_amd.build` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SplitVaeError(
            "libsplitvae_hip.so not found at %s -- the HIP extension is required (no CPU fallback). "
            "Build it: python split_vae_amd/build.py" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SplitVaeError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the ABI drifted from the header
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise SplitVaeError("%s failed: %s" % (what, STATUS.get(rc, str(rc))))
    raise SplitVaeError("%s failed: hipError_t %d" % (what, rc))
