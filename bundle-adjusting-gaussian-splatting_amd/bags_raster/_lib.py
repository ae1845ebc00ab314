"""ctypes binding of libbags_raster.so (include/bags_raster.h).  No torch types cross this boundary: only raw
device pointers (``tensor.data_ptr()``), ints and floats.  The library is built in-tree by ``__graft_entry__.build()``
(or ``make -C csrc``); importing this module without it raises -- there is no CPU or eager fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BAGS_RASTER_LIB: another build of the SAME library (csrc/Makefile `asan`: host-side AddressSanitizer build for the CPU ABI
# tests).  Not a fallback: a missing file still raises.
LIB_PATH = os.environ.get("BAGS_RASTER_LIB") or os.path.join(_HERE, "libbags_raster.so")

ABI_VERSION = 10
TILES_AABB, TILES_OPACITY = 0, 1
DEPTH_Z, DEPTH_DISTANCE = 0, 1
BINNING_AUTO, BINNING_RADIX = 0, 1
CLAMP_GRAD_STOCK, CLAMP_GRAD_EXACT = 0, 1
CONIC_GRAD_STOCK, CONIC_GRAD_EXACT = 0, 1
BWD_ALL, BWD_BLEND, BWD_PREPROCESS = 0, 1, 2

c_fp = C.c_void_p  # device pointers travel as integers


class BagsSettings(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
                ("depth_key", C.c_int32), ("debug", C.c_int32), ("debug_iter", C.c_int32),
                ("tile_bounds", C.c_int32), ("binning", C.c_int32), ("clamp_grad", C.c_int32), ("conic_grad", C.c_int32),
                ("bg", c_fp), ("viewmatrix", c_fp), ("projmatrix", c_fp), ("intrinsic", c_fp), ("campos", c_fp)]


class BagsInputs(C.Structure):
    _fields_ = [("P", C.c_int32), ("means3D", c_fp), ("means2D", c_fp), ("shift_factors", c_fp), ("shs", c_fp),
                ("colors_precomp", c_fp), ("opacities", c_fp), ("scales", c_fp), ("rotations", c_fp),
                ("cov3D_precomp", c_fp), ("shs_rest", c_fp)]


class BagsState(C.Structure):
    _fields_ = [("geom", c_fp), ("geom_bytes", C.c_size_t), ("binning", c_fp), ("binning_bytes", C.c_size_t),
                ("image", c_fp), ("image_bytes", C.c_size_t)]


class BagsForwardOut(C.Structure):
    _fields_ = [("color", c_fp), ("radii", c_fp), ("depth", c_fp), ("weights", c_fp), ("mean2D", c_fp)]


class BagsBackwardArgs(C.Structure):
    _fields_ = [("grad_color", c_fp), ("num_rendered", C.c_int64), ("workspace", c_fp), ("workspace_bytes", C.c_size_t),
                ("grad_means3D", c_fp), ("grad_means2D", c_fp), ("grad_means2D_densify", c_fp), ("grad_shs", c_fp),
                ("grad_colors_precomp", c_fp), ("grad_opacities", c_fp), ("grad_scales", c_fp), ("grad_rotations", c_fp),
                ("grad_cov3D_precomp", c_fp), ("grad_viewmatrix", c_fp), ("grad_projmatrix", c_fp),
                ("grad_intrinsic", c_fp), ("grad_campos", c_fp), ("grad_shift_factors", c_fp),
                ("binning_capacity", C.c_int64), ("accumulate", C.c_int32), ("dense_per_tile", C.c_int32), ("grad_shs_rest", c_fp),
                ("phase", C.c_int32), ("reserved2", C.c_int32), ("grad_dldc", c_fp)]


class BagsDebugViews(C.Structure):
    _fields_ = [("tiles_touched", c_fp), ("rect", c_fp), ("depth_bits", c_fp), ("point_list", c_fp),
                ("keys_sorted", c_fp), ("ranges", c_fp), ("n_contrib", c_fp), ("final_T", c_fp)]


MAX_SH_VIEWS = 16


class BagsShViews(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("reserved", C.c_int32), ("campos", c_fp * MAX_SH_VIEWS), ("dldc", c_fp * MAX_SH_VIEWS)]


# every symbol include/bags_raster.h declares: (restype, argtypes)
class BagsCamera(C.Structure):
    _fields_ = [("init_quaternion", C.c_void_p), ("delta_quaternion", C.c_void_p), ("init_translation", C.c_void_p),
                ("delta_translation", C.c_void_p), ("fovx", C.c_void_p), ("fovy", C.c_void_p),
                ("global_rotation", C.c_void_p), ("global_translation_scale", C.c_void_p),
                ("znear", C.c_float), ("zfar", C.c_float)]


class BagsRawGaussians(C.Structure):
    _fields_ = [("P", C.c_int32), ("K", C.c_int32), ("features_dc", C.c_void_p), ("features_rest", C.c_void_p),
                ("opacity", C.c_void_p), ("scaling", C.c_void_p), ("rotation", C.c_void_p)]


SYMBOLS = {
    "bags_abi_version": (C.c_int, []),
    "bags_build_info": (C.c_char_p, []),
    "bags_last_error": (C.c_char_p, []),
    "bags_geom_size": (C.c_size_t, [C.c_int32]),
    "bags_binning_size": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "bags_image_size": (C.c_size_t, [C.c_int32, C.c_int32]),
    "bags_backward_workspace_size": (C.c_size_t, [C.c_int32, C.c_int64]),
    "bags_forward_prepare": (C.c_int, [C.POINTER(BagsSettings), C.POINTER(BagsInputs), C.POINTER(BagsState),
                                       C.POINTER(BagsForwardOut), C.POINTER(C.c_int64), C.c_void_p]),
    "bags_forward_finish": (C.c_int, [C.POINTER(BagsSettings), C.POINTER(BagsInputs), C.POINTER(BagsState),
                                      C.POINTER(BagsForwardOut), C.c_int64, C.c_void_p]),
    "bags_forward_prepare_async": (C.c_int, [C.POINTER(BagsSettings), C.POINTER(BagsInputs), C.POINTER(BagsState),
                                             C.POINTER(BagsForwardOut), C.c_void_p, C.c_void_p]),
    "bags_forward_finish_speculative": (C.c_int, [C.POINTER(BagsSettings), C.POINTER(BagsInputs), C.POINTER(BagsState),
                                                  C.POINTER(BagsForwardOut), C.c_int64, C.c_void_p]),
    "bags_backward": (C.c_int, [C.POINTER(BagsSettings), C.POINTER(BagsInputs), C.POINTER(BagsState),
                                C.POINTER(BagsBackwardArgs), C.c_void_p]),
    "bags_sh_gradient_from_views": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(BagsShViews), C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p]),
    "bags_debug_views": (C.c_int, [C.POINTER(BagsSettings), C.POINTER(BagsInputs), C.POINTER(BagsState), C.c_int64,
                                   C.POINTER(BagsDebugViews), C.c_void_p]),
    "bags_profile_enable": (C.c_int, [C.c_int]),
    "bags_profile_stride": (C.c_int, [C.c_int]),
    "bags_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "bags_loss_workspace_size": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "bags_loss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p]),
    "bags_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "bags_photometric_loss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                                C.c_float, C.c_void_p, C.c_void_p]),
    "bags_photometric_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                                 C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bags_camera_forward": (C.c_int, [C.POINTER(BagsCamera), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bags_camera_backward": (C.c_int, [C.POINTER(BagsCamera)] + [C.c_void_p] * 11),
    "bags_resample_forward": (C.c_int, [C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p] * 4),
    "bags_resample_workspace_size": (C.c_size_t, [C.c_int32] * 4),
    "bags_resample_backward": (C.c_int, [C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p] + [C.c_int32] * 6 +
                               [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bags_activations_forward": (C.c_int, [C.POINTER(BagsRawGaussians)] + [C.c_void_p] * 5),
    "bags_activations_backward": (C.c_int, [C.POINTER(BagsRawGaussians)] + [C.c_void_p] * 10),
    "bags_knn_workspace_size": (C.c_size_t, [C.c_int32]),
    "bags_knn_mean_dist2": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "bags_compute_relocation": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int32, C.c_int32, c_fp, c_fp, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the HIP library (once).  Raises if it is absent or its ABI does not match."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"bags_raster: {LIB_PATH} is missing. Build the HIP extension first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or make -C csrc). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.bags_abi_version() != ABI_VERSION:
        raise ImportError(f"bags_raster: ABI {lib.bags_abi_version()} != expected {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().bags_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg} (code {rc})")


def profile_enable(mode) -> None:
    """0/False off, 1 dominant kernel only, 2/True every stage."""
    load().bags_profile_enable(2 if mode is True else int(mode))


def profile_stride(n: int) -> None:
    """mode 1: only every n-th launch of the dominant kernel carries timing events."""
    load().bags_profile_stride(int(n))


def profile_read():
    """{stage: (total_ms, intervals)} since the last read; synchronises on the recorded events."""
    n = 16
    names = (C.c_char_p * n)()
    ms = (C.c_double * n)()
    calls = (C.c_int64 * n)()
    k = load().bags_profile_read(n, names, ms, calls)
    return {names[i].decode(): (ms[i], calls[i]) for i in range(min(k, n))}
