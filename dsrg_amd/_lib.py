"""ctypes binding of the C ABI in include/dsrg_hip.h (libdsrg_hip.so).

There is deliberately no fallback: if the shared library is missing or a call
fails, an exception is raised.  The product never routes through the CPU oracle.
"""
import ctypes
import os

import torch  # noqa: F401  — loaded first so libdsrg_hip.so binds to torch's libamdhip64.so.7

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSRG_LIB (tools only): another build of the same library next to the shipped one, e.g. libdsrg_hip.exp3.so from
# `make -C dsrg_amd/csrc EXP=3 exp` — for A/B measurements of kernel variants on one box
LIB_PATH = os.path.join(_HERE, os.environ.get("DSRG_LIB") or "libdsrg_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = 0, -1, -2, -3, -4


class DsrgError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libdsrg_hip error %d: %s" % (code, msg))
        self.code = code


class CrfParams(ctypes.Structure):
    """dsrg_crf_params (include/dsrg_hip.h) — the arguments of krahenbuhl2013.CRF (CRF.py:25-35)."""
    _fields_ = [("w_bilateral", ctypes.c_float), ("theta_alpha_x", ctypes.c_float),
                ("theta_alpha_y", ctypes.c_float), ("theta_beta_r", ctypes.c_float),
                ("theta_beta_g", ctypes.c_float), ("theta_beta_b", ctypes.c_float),
                ("w_gaussian", ctypes.c_float), ("theta_gamma_x", ctypes.c_float),
                ("theta_gamma_y", ctypes.c_float), ("n_iters", ctypes.c_int32)]

    @classmethod
    def from_crf_args(cls, maxiter=10, scale_factor=1.0, color_factor=13):
        # same Python-double arithmetic as CRF.py:31-32, rounded to float at the C boundary
        return cls(10.0, 80 / scale_factor, 80 / scale_factor, color_factor, color_factor, color_factor,
                   3.0, 3 / scale_factor, 3 / scale_factor, int(maxiter))


_vp, _i, _f, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/dsrg_hip.h declares
SIGNATURES = {
    "dsrg_last_error": (ctypes.c_char_p, []),
    "dsrg_device_count": (_i, []),
    "dsrg_host_register": (_i, [_vp, _sz]),
    "dsrg_host_unregister": (_i, [_vp]),
    "dsrg_crf_create": (_i, [_i, _i, _i, ctypes.POINTER(_vp)]),
    "dsrg_crf_create_batch": (_i, [_i, _i, _i, _i, ctypes.POINTER(_vp)]),
    "dsrg_crf_destroy": (_i, [_vp]),
    "dsrg_crf_set_unary_energy": (_i, [_vp, _vp]),
    "dsrg_crf_add_pairwise_energy": (_i, [_vp] + [_f] * 9 + [_vp]),
    "dsrg_crf_inference": (_i, [_vp, _i, _vp]),
    "dsrg_crf_map": (_i, [_vp, _i, _vp]),
    "dsrg_crf_set_stream": (_i, [_vp, _vp, _i]),
    "dsrg_crf_synchronize": (_i, [_vp]),
    "dsrg_crf_npixels": (_i, [_vp]),
    "dsrg_crf_nlabels": (_i, [_vp]),
    "dsrg_crf_lattice_size": (_i, [_vp, _i]),
    "dsrg_crf_profile_start": (_i, [_vp, _i]),
    "dsrg_crf_profile_stop": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    "dsrg_ctx_create": (_i, [_i, _i, _i, _i, ctypes.POINTER(_vp)]),
    "dsrg_ctx_destroy": (_i, [_vp]),
    "dsrg_crf_refine_batch": (_i, [_vp, _i, _vp, _vp, _i, _i, ctypes.POINTER(CrfParams), _vp, _vp, _vp]),
    "dsrg_crf_prepare_batch": (_i, [_vp, _i, _vp, _i, _i, ctypes.POINTER(CrfParams), _vp]),
    "dsrg_crf_meanfield_batch": (_i, [_vp, _i, _vp, _vp, ctypes.POINTER(CrfParams), _vp, _vp]),
    "dsrg_ctx_lattice_sizes": (_i, [_vp, _i, _vp, _vp, _vp]),
    "dsrg_ctx_lattice_extras": (_i, [_vp, _i, _vp, _vp, _vp]),
    "dsrg_ctx_filter_plan": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "dsrg_ctx_lattice_dump": (_i, [_vp, _i, _i, ctypes.POINTER(ctypes.c_int32), _vp, _vp, _vp, _vp, _vp, _vp]),
    "dsrg_ctx_read_refined": (_i, [_vp, _i, _vp, _vp]),
    "dsrg_ctx_lattice_norm": (_i, [_vp, _i, _i, _vp, _vp]),
    "dsrg_ctx_filter_once": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "dsrg_ctx_profile_start": (_i, [_vp, _i]),
    "dsrg_ctx_profile_stop": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    "dsrg_crf_layer_backward": (_i, [_sz, _vp, _vp, _vp, _vp]),
    "dsrg_srg_grow_batch": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp]),
    "dsrg_softmax_forward": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "dsrg_softmax_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dsrg_seed_loss": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dsrg_constrain_loss": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dsrg_seed_loss_plain": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dsrg_expand_loss": (_i, [_i, _i, _i, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp]),
    "dsrg_confusion_matrix": (_i, [_sz, _vp, _vp, _i, _i, _vp, _vp]),
    "dsrg_im2col3x3_nhwc16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dsrg_relu_bwd_bias_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, ctypes.c_long, _i, ctypes.c_float, _vp]),
    "dsrg_col2im3x3_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dsrg_avgpool3x3_s1_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dsrg_conv_igemm_backward_residual_bf16": (_i, [_vp] * 8 + [_i, _vp, _sz] + [_i] * 6 + [_vp]),
    "dsrg_pack_conv_weight_scaled_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "dsrg_aspp_shift_sum_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 6 + [_vp]),
    "dsrg_aspp_shift_gather_bf16": (_i, [_vp, _vp, _vp] + [_i] * 6 + [_vp]),
    "dsrg_defer_reductions": (_i, [_i]),
    "dsrg_flush_reductions": (_i, [_vp]),
    "dsrg_conv_igemm_residual_bf16": (_i, [_vp] * 6 + [_i] * 8 + [_vp]),
    "dsrg_conv_igemm_split_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "dsrg_add_relu_bf16": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "dsrg_relu_mask_bf16": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "dsrg_bias_grad_bf16": (_i, [_vp, _vp, _vp, _i, ctypes.c_long, _i, _vp]),
    "dsrg_heads_forward_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dsrg_conv3x3_direct_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_conv3x3_direct_dgrad_workspace": (_sz, [_i]),
    "dsrg_conv3x3_direct_dgrad_bf16": (_i, [_vp] * 6 + [_sz] + [_i] * 5 + [_vp]),
    "dsrg_conv_igemm_supported": (_i, [_i, _i, _i]),
    "dsrg_conv_igemm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, ctypes.c_ulonglong, _vp, _sz,
                                  _vp]),
    "dsrg_conv_igemm_workspace": (_sz, []),
    "dsrg_conv_igemm_dgrad_workspace": (_sz, [_i] * 5),
    "dsrg_conv_igemm_dgrad_bf16": (_i, [_vp] * 6 + [_i] * 7 + [_f, _vp, _sz, _vp]),
    "dsrg_conv_igemm_workspace_status": (_i, [_vp, _vp, _vp]),
    "dsrg_pack_conv_weight_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "dsrg_sgd_pack_f32": (_i, [_i] + [_vp] * 9 + [_f, _vp]),
    "dsrg_conv_igemm_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "dsrg_conv_igemm_wgrad_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_conv_igemm_backward_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_conv3x3_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "dsrg_conv3x3_wgrad_bf16": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "dsrg_conv3x3_wgrad_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "dsrg_pack_conv_weight_direct_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "dsrg_heads_backward_chunks": (_i, [_i]),
    "dsrg_heads_backward_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dsrg_heads_backward_relu_workspace": (_sz, [_i] * 3),
    "dsrg_heads_backward_relu_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp, _vp, _f, _vp, _vp, _sz] + [_i] * 4 + [_vp]),
    "dsrg_maxpool3x3_fwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_maxpool3x3_relu_fwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_maxpool3x3_bwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_maxpool3x3_bwd_relu_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dsrg_supervision_step": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _d, _d, ctypes.POINTER(CrfParams),
                                   _vp, _vp, _vp, _vp, _vp, _vp]),
}

_LIB = None


def lib():
    """Load libdsrg_hip.so (built by dsrg_amd/csrc/Makefile or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing — build it with `make -C dsrg_amd/csrc` (hipcc --offload-arch=gfx950) or "
                "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        raise DsrgError(rc, lib().dsrg_last_error().decode("utf-8", "replace"))


def require_gpu():
    if lib().dsrg_device_count() < 1:
        raise DsrgError(ERR_HIP, "no HIP device visible — the DSRG supervision kernels need an MI355X "
                                 "(gfx950); there is no CPU fallback")
