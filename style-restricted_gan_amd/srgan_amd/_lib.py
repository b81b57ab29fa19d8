"""ctypes binding of libsrgan_hip.so (C ABI declared in include/srgan_hip.h).

There is NO fallback: if the library is missing or an entry point fails, the error is raised.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRGAN_HIP_LIB") or os.path.join(_HERE, "libsrgan_hip.so")   # override: A/B kernel experiments
# An experiment build (csrc/Makefile `make exp`, scratch/libsrgan_exp.so) is loaded: only then are the A/B switches of the
# measurement scripts honoured, here (ab()) and in the library (SRGAN_AB_SET in csrc/common.h).  bench.py reports it.
EXPERIMENTS = bool(os.environ.get("SRGAN_HIP_LIB"))


def ab(name):
    """An A/B switch of scratch/: the environment variable `name`, read only next to an experiment build of the library."""
    return EXPERIMENTS and bool(os.environ.get(name))


def active_switches():
    """What makes this process differ from the product configuration (bench.py prints it and refuses to call the line a result)."""
    out = {}
    if EXPERIMENTS:
        out["SRGAN_HIP_LIB"] = os.environ["SRGAN_HIP_LIB"]
        out.update({k: v for k, v in os.environ.items() if k.startswith("SRGAN_") and k != "SRGAN_HIP_LIB"})
    return out


class ConvDesc(ctypes.Structure):
    """struct srgan_conv_desc"""
    _fields_ = [("N", c_int), ("Hi", c_int), ("Wi", c_int), ("I", c_int),
                ("Ho", c_int), ("Wo", c_int), ("O", c_int),
                ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int),
                ("pad_mode", c_int),
                ("sO", c_longlong), ("sI", c_longlong), ("sH", c_longlong), ("sW", c_longlong)]


P = c_void_p
_DESC = POINTER(ConvDesc)

# name -> (restype, argtypes); must list EVERY symbol include/srgan_hip.h declares
SIGNATURES = {
    "srgan_abi_version": (c_int, []),
    "srgan_comm_available": (c_int, []),
    "srgan_comm_unique_id": (c_int, [P]),
    "srgan_comm_init": (c_int, [P, c_int, c_int, POINTER(c_void_p)]),
    "srgan_comm_size": (c_int, [P, POINTER(c_int)]),
    "srgan_comm_destroy": (c_int, [P]),
    "srgan_allreduce_bucket": (c_int, [P, P, c_longlong, c_int, c_int, P]),
    "srgan_allgather_rows": (c_int, [P, P, P, c_longlong, P]),
    "srgan_last_error": (ctypes.c_char_p, []),
    "srgan_conv2d_workspace": (c_size_t, [_DESC]),
    "srgan_conv2d_fwd": (c_int, [_DESC, P, P, P, P, c_int, c_float, P, c_size_t, P]),
    "srgan_conv2d_dgrad": (c_int, [_DESC, P, P, P, P, c_size_t, P]),
    "srgan_conv2d_packed_bytes": (c_size_t, [_DESC, c_int, c_int]),
    "srgan_conv2d_pack": (c_int, [_DESC, c_int, c_int, P, P, c_size_t, P]),
    "srgan_conv2d_packed_scratch": (c_size_t, [_DESC, c_int]),
    "srgan_conv2d_pack_signature": (ctypes.c_ulonglong, [_DESC, c_int, c_int]),
    "srgan_conv2d_fwd_packed": (c_int, [_DESC, P, P, P, P, c_int, c_float, P, c_size_t, P]),
    "srgan_conv2d_dgrad_packed": (c_int, [_DESC, P, P, P, P, c_size_t, P]),
    "srgan_conv2d_dgrad_packed_mask": (c_int, [_DESC, P, P, P, c_float, P, P, c_size_t, P]),
    "srgan_conv2d_dgrad_packed_add": (c_int, [_DESC, P, P, P, P, P, c_size_t, P]),
    "srgan_instnorm_conv_v_applicable": (c_int, [_DESC]),
    "srgan_instnorm_fwd_v": (c_int, [_DESC, P, P, P, P, P, P, c_size_t, c_float, c_int, c_float, P]),
    "srgan_conv2d_fwd_from_v": (c_int, [_DESC, P, P, P, P, c_int, c_float, P]),
    "srgan_instnorm_bwd_vz_applicable": (c_int, [_DESC]),
    "srgan_instnorm_bwd_vz_z_bytes": (c_size_t, [_DESC]),
    "srgan_instnorm_bwd_vz": (c_int, [_DESC, P, P, P, P, P, P, P, P, P, c_size_t, P, c_size_t, c_int, c_float, P]),
    "srgan_conv2d_dgrad_from_v": (c_int, [_DESC, P, P, P, P, P]),
    "srgan_conv2d_wgrad_vz": (c_int, [_DESC, P, P, P, P, c_size_t, P]),
    "srgan_halo16_applicable": (c_int, [_DESC]),
    "srgan_halo16s2_applicable": (c_int, [_DESC]),
    "srgan_instnorm_io_applicable": (c_int, [c_int, c_int, c_int]),
    "srgan_instnorm_fwd_io": (c_int, [P, c_int, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_float, c_int, c_float, P, c_size_t, P]),
    "srgan_instnorm_bwd_io": (c_int, [P, c_int, P, c_int, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "srgan_halo16_conv": (c_int, [_DESC, c_int, P, c_int, P, P, P, c_int, P]),
    "srgan_halo16_wgrad": (c_int, [_DESC, P, c_int, P, c_int, P, P, c_size_t, P]),
    "srgan_conv2d_io_applicable": (c_int, [_DESC, c_int]),
    "srgan_conv2d_io_fwd": (c_int, [_DESC, P, c_int, P, P, P, c_int, c_int, c_float, P, c_size_t, P]),
    "srgan_conv2d_io_dgrad": (c_int, [_DESC, P, c_int, P, P, c_int, P, c_size_t, P]),
    "srgan_act_bwd_io": (c_int, [P, c_int, P, c_int, P, c_int, c_longlong, c_int, c_float, P]),
    "srgan_igemm16_io_applicable": (c_int, [_DESC, c_int]),
    "srgan_igemm16_conv": (c_int, [_DESC, c_int, P, c_int, P, P, P, c_int, c_int, c_float, P, c_size_t, P]),
    "srgan_igemm16_wgrad": (c_int, [_DESC, P, P, P, P, c_size_t, P]),
    "srgan_instnorm_slab_applicable": (c_int, [c_int, c_int, c_int]),
    "srgan_instnorm_slab_fwd_io": (c_int, [P, c_int, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_float, c_int, c_float, P]),
    "srgan_instnorm_slab_bwd_io": (c_int, [P, c_int, P, c_int, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "srgan_set_wgrad_accumulate": (c_int, [c_int]),
    "srgan_wgrad_defer_begin": (c_int, [P, c_size_t, P]),
    "srgan_wgrad_defer_end": (c_int, []),
    "srgan_wgrad_defer_stats": (c_int, [POINTER(c_longlong), POINTER(c_longlong)]),
    "srgan_wgrad_defer_need": (c_int, [POINTER(c_longlong)]),
    "srgan_conv2d_wgrad": (c_int, [_DESC, P, P, P, P, P, c_size_t, P]),
    "srgan_conv2d_wgrad_v_bytes": (c_size_t, [_DESC]),
    "srgan_conv2d_wgrad_v": (c_int, [_DESC, P, P, P, P, P, c_size_t, P]),
    "srgan_instnorm_workspace": (c_size_t, [c_int, c_int, c_int]),
    "srgan_instnorm_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, c_float, P, c_size_t, P]),
    "srgan_instnorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "srgan_cbin_affine_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "srgan_cbin_affine_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    "srgan_cbin_rec_bytes": (c_size_t, []),
    "srgan_cbin_rec_fill": (c_int, [P] * 15 + [c_int]),
    "srgan_cbin_rec_set_accumulate": (c_int, [P, c_int]),
    "srgan_cbin_affine_multi_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_cbin_affine_multi_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "srgan_act_fwd": (c_int, [P, P, c_longlong, c_int, c_float, P]),
    "srgan_act_bwd": (c_int, [P, P, P, c_longlong, c_int, c_float, P]),
    "srgan_tanh_fwd": (c_int, [P, P, c_longlong, P]),
    "srgan_tanh_bwd": (c_int, [P, P, P, c_longlong, P]),
    "srgan_add": (c_int, [P, P, P, c_longlong, P]),
    "srgan_avgpool3s2_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_avgpool3s2_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_avgpool2_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_avgpool2_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_avgpool2_fwd_io": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "srgan_avgpool2_bwd_io": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "srgan_lrelu_gap_fwd": (c_int, [P, P, c_int, c_int, c_int, c_float, P]),
    "srgan_lrelu_gap_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P]),
    "srgan_reparam_fwd": (c_int, [P, P, P, P, P, c_longlong, P]),
    "srgan_reparam_bwd": (c_int, [P, P, P, P, c_longlong, P]),
    "srgan_linear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "srgan_linear_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "srgan_nchw_to_nhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_nhwc_to_nchw": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_mse_const": (c_int, [P, c_longlong, c_float, c_float, P, P, P]),
    "srgan_softmax_mse": (c_int, [P, P, c_int, c_int, c_float, P, P, P, P]),
    "srgan_softmax_xent": (c_int, [P, P, c_int, c_int, c_float, P, P, P]),
    "srgan_l1_workspace": (c_size_t, [c_longlong]),
    "srgan_l1_mean": (c_int, [P, P, c_longlong, c_float, P, P, P, P, c_size_t, P]),
    "srgan_latent_losses": (c_int, [P, c_int, c_int, c_float, P, c_int, c_float, c_float, c_float, c_float, c_float, P, P, P, P]),
    "srgan_mse_pair": (c_int, [P, P, c_longlong, c_float, P, P, P, P]),
    "srgan_kl_normal": (c_int, [P, P, c_longlong, c_float, P, P, P, P]),
    "srgan_d_losses": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, c_float, c_float, c_float, P, P, P, P]),
    "srgan_lincomb": (c_int, [P, P, c_int, P, P]),
    "srgan_lincomb_bwd": (c_int, [P, c_int, P, P, P]),
    "srgan_soft_histogram_workspace": (c_size_t, [c_longlong, c_int]),
    "srgan_soft_histogram_fwd": (c_int, [P, c_longlong, c_int, c_float, c_float, c_float, P, P, c_size_t, P]),
    "srgan_soft_histogram_bwd": (c_int, [P, P, c_longlong, c_int, c_float, c_float, c_float, P, P]),
    "srgan_adam_multi": (c_int, [P, c_int, c_longlong, c_float, c_float, c_float, c_float, c_int, P]),
    "srgan_pack_entry_bytes": (c_size_t, []),
    "srgan_conv2d_pack_entry": (c_int, [POINTER(ConvDesc), c_int, c_int, P, P, P]),
    "srgan_conv2d_pack_multi": (c_int, [P, c_int, P]),
    "srgan_set_compute_mode": (c_int, [c_int]),
    "srgan_get_compute_mode": (c_int, []),
    "srgan_preprocess_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "srgan_preprocess_u8": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, c_int, P,
                                    c_int, c_int, P, P, c_size_t, P]),
    "srgan_prof_enable": (c_int, [c_int]),
    "srgan_prof_num_kernels": (c_int, []),
    "srgan_prof_kernel_name": (ctypes.c_char_p, [c_int]),
    "srgan_prof_collect": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_longlong), POINTER(ctypes.c_double)]),
    "srgan_prof_num_slots": (c_int, []),
    "srgan_prof_slot": (c_int, [c_int, POINTER(c_int), POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "srgan_adam_step": (c_int, [P, P, P, P, c_longlong, c_float, c_float, c_float, c_float, c_int, P]),
    "srgan_adam_state_bytes": (c_size_t, []),
    "srgan_adam_state_init": (c_int, [P, c_float, c_float, c_float, c_float, c_int, P]),
    "srgan_adam_state_set_lr": (c_int, [P, c_float, P]),
    "srgan_adam_multi_dev": (c_int, [P, c_int, c_longlong, P, P]),
    "srgan_upload_small": (c_int, [P, P, c_size_t, P]),
    "srgan_maxpool2_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "srgan_pairwise_dist": (c_int, [P, c_int, P, c_int, c_int, P, P]),
    "srgan_kth_smallest_rows": (c_int, [P, c_int, c_int, c_int, P, P]),
    "srgan_prdc_workspace": (c_size_t, [c_int, c_int]),
    "srgan_prdc_from_dist": (c_int, [P, c_int, c_int, P, P, c_int, P, P, c_size_t, P]),
}

_lib = None


class SrganHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrganHipError(
            f"{LIB_PATH} not found: build it with `make -C style-restricted_gan_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.srgan_abi_version() != 1:
        raise SrganHipError("libsrgan_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().srgan_last_error().decode(errors="replace")
        raise SrganHipError(f"{what} failed (code {code}): {msg}")
