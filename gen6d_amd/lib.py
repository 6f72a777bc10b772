"""ctypes binding of libgen6d_hip.so (include/gen6d_hip.h).  Fails loudly when the library is missing: there is no
CPU or PyTorch fallback for the kernels behind this ABI."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgen6d_hip.so")      # (tools/ load profiling builds by assigning lib.LIB_PATH before the first call)

G6D_ERRORS = {-1: "G6D_EINVAL", -2: "G6D_ENOSPC", -3: "G6D_ELAUNCH"}


class G6dWinoSeg(C.Structure):
    """include/gen6d_hip.h: one map size of g6d_wino_conv3x3_multi."""
    _fields_ = [("in_", C.c_void_p), ("out_full", C.c_void_p), ("out_pool", C.c_void_p),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("ld_in", C.c_int32), ("ld_full", C.c_int32), ("ld_pool", C.c_int32)]


class G6dConv16Seg(C.Structure):
    """include/gen6d_hip.h: one map size of g6d_conv16_direct_multi (16-bit activations)."""
    _fields_ = [("in_", C.c_void_p), ("out_full", C.c_void_p), ("out_pool", C.c_void_p),
                ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("ld_in", C.c_int32), ("ld_full", C.c_int32),
                ("ld_pool", C.c_int32)]


class G6dCorrSeg(C.Structure):
    """include/gen6d_hip.h: one map of g6d_corr2d_patch_multi."""
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("ld_in", C.c_int32), ("ld_out", C.c_int32),
                ("N", C.c_int32), ("reserved_", C.c_int32)]


class G6dConv(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("mul", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("stats", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("N", C.c_int32), ("Di", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32), ("ld_in", C.c_int32),
        ("Do", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("ld_out", C.c_int32),
        ("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
        ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
        ("in_relu", C.c_int32), ("in_affine_per_n", C.c_int32), ("out_act", C.c_int32),
        ("stat_rows_per_group", C.c_int32), ("split_k", C.c_int32), ("math_mode", C.c_int32),
        ("weight_wino", C.c_void_p),
        ("fin_scale", C.c_void_p), ("fin_shift", C.c_void_p), ("fin_counter", C.c_void_p),
        ("fin_count", C.c_double), ("fin_eps", C.c_double), ("fin_groups", C.c_int32),
        ("in_image_mod", C.c_int32), ("mul_group_images", C.c_int32), ("reserved_", C.c_int32),
        ("weight_wino16", C.c_void_p), ("weight_wino43", C.c_void_p),
    ]


_P, _I, _F, _D = C.c_void_p, C.c_int, C.c_float, C.c_double

# name -> argtypes (every function returns int); mirrors include/gen6d_hip.h
SIGNATURES = {
    "g6d_marker": [_I, _P],
    "g6d_conv_igemm": [C.POINTER(G6dConv), _P],
    "g6d_conv_plan": [C.POINTER(G6dConv)],
    "g6d_corr2d_patch": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _I, _P, C.c_size_t, _I, _P],
    "g6d_corr2d_patch_multi": [_P, _I, _I, _P, _I, _I, _I, _P, C.c_size_t, _I, _P],
    "g6d_corr2d_wino_multi": [_P, _I, _I, _P, _I, _I, _P, C.c_size_t, _P],
    "g6d_corr2d_patch16_multi": [_P, _I, _I, _P, _I, _I, _I, _P, C.c_size_t, _I, _P],
    "g6d_stats_finalize": [_P, _I, _D, _D, _P, _P, _P],
    "g6d_affine_act_pool": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "g6d_upsample_bilinear": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "g6d_bias_relu_pool_nchw": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "g6d_vgg_conv1_pool": [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P],
    "g6d_vgg_conv1_pool_nhwc": [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P],
    "g6d_vgg_conv1_pool_nhwc16": [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _I, _P],
    "g6d_conv16_direct_multi": [_P, _I, _I, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "g6d_corr16_multi": [_P, _I, _I, _P, _F, _I, _I, _I, _P],
    "g6d_product_split16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "g6d_affine_split16": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "g6d_vgg_conv1_pool_nhwc_norm": [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P],
    "g6d_wino_conv3x3": [_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _I, _P, C.c_size_t, _P],
    "g6d_wino_conv3x3_multi": [_P, _I, _I, _P, _P, _I, _I, _P, C.c_size_t, _P],
    "g6d_wino16_conv3x3_multi": [_P, _I, _I, _P, _P, _I, _I, _I, _P, C.c_size_t, _P],
    "g6d_wino43_conv3x3_multi": [_P, _I, _I, _P, _P, _I, _I, _P, C.c_size_t, _P],
    "g6d_corr2d_wino43_multi": [_P, _I, _I, _P, _I, _I, _P, C.c_size_t, _P],
    "g6d_l2norm_rows": [_P, _I, _I, _I, _P],
    "g6d_nchw_to_nhwc": [_P, _I, _I, _I, _I, _I, _P, _I, _P],
    "g6d_selector_ref_sums": [_P, _I, _I, _I, _P, _P, _P],
    "g6d_selector_prod_affine": [_P, _P, _P, _I, _I, _I, _D, _P, _P, _P],
    "g6d_selector_scan": [_P, _P, _I, _I, _I, _P, _P, _P],
    "g6d_selector_levels": [_I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _D, _P, _P, _P, _P, _P],
    "g6d_refiner_volume": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "g6d_refiner_volume_kp": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "g6d_detector_assemble": [_P, _P, _P, _I, _I, _I, C.POINTER(C.c_float), _F, _I, _I, _I, _I, _P, _I, _P],
    "g6d_detector_score_mlp_max": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "g6d_detector_decode": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P],
    "g6d_resize_bilinear_pyramid": [_P, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), _P],
    "g6d_zero_bytes": [_P, C.c_size_t, _P],
    "g6d_vps_norm": [_P, _I, _P, _I, _I, _I, _P],
    "g6d_max_an_add": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P],
    "g6d_attention": [_P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P],
    "g6d_layernorm": [_P, _I, _I, _I, _P, _P, _F, _P, _I, _P],
    "g6d_affine_act_add": [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P],
    "g6d_linear_gemv": [_P, _I, _I, _P, _P, _I, _I, _P, _P],
    "g6d_linear_gemv_batch": [_P, _I, _I, _P, _P, _I, _I, _P, _P, C.c_size_t, _P],
    "g6d_chain_crop_from_detection": [_P, _F, _P, _I, _P],
    "g6d_chain_pose_from_selection": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "g6d_chain_refine_prepare": [_P, _P, _P, _F, _F, _P, _P, _I, _I, _P, _P, _F, _P, _I, _P],
    "g6d_chain_refine_update": [_P, _P, _P, _P, _I, _P, _P, _I, _P],
    "g6d_warp_batch": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _P],
    "g6d_warp_perspective": [_P, _I, _I, _I, C.POINTER(C.c_float), _P, _I, _I, _I, _F, _P],
}

_lib = None


def load():
    """Load (once) and type the shared library. Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           "(make -C gen6d_amd/csrc). There is no fallback path.")
    # PyTorch-ROCm ships its own HIP runtime: it has to be in the process BEFORE this library is loaded, so that both resolve to the
    # same libamdhip64 (loaded the other way round the launches here see "no ROCm-capable device")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.g6d_abi_version.restype = C.c_int
    lib.g6d_sizeof_conv_desc.restype = C.c_int
    if lib.g6d_sizeof_conv_desc() != C.sizeof(G6dConv):
        raise RuntimeError("libgen6d_hip.so: G6dConv layout differs from the ctypes binding (stale build?)")
    lib.g6d_last_error.restype = C.c_char_p
    lib.g6d_set_knob.argtypes, lib.g6d_set_knob.restype = [C.c_char_p, C.c_double], C.c_int
    lib.g6d_get_knob.argtypes, lib.g6d_get_knob.restype = [C.c_char_p], C.c_double
    lib.g6d_reset_knobs.argtypes, lib.g6d_reset_knobs.restype = [], None
    _lib = lib
    return lib


def set_knob(name, value):
    """Launch-policy knob of the library (include/gen6d_hip.h): tools/ and tests force kernel variants / sweep model constants with it."""
    check(load().g6d_set_knob(name.encode(), float(value)), f"g6d_set_knob({name})")


def get_knob(name):
    return load().g6d_get_knob(name.encode())


def reset_knobs():
    load().g6d_reset_knobs()


def check(rc, what):
    if rc != 0:
        msg = load().g6d_last_error().decode()
        raise RuntimeError(f"{what} failed: {G6D_ERRORS.get(rc, rc)} ({msg})")
