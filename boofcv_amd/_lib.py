"""ctypes binding of libboofhip.so -- one entry per function declared in include/boofhip.h.

The library is the product; there is no Python or CPU fallback.  Loading fails loudly when the shared object is missing
(run `python -m boofcv_amd.build` or `__graft_entry__.build()`), and creating a context fails loudly without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BHIP_LIB selects another build of the same ABI (scripts/ use libboofhip_exp.so, the -DBHIP_EXPERIMENTS build, for ablation runs)
LIB_PATH = os.environ.get("BHIP_LIB") or os.path.join(_HERE, "libboofhip.so")

BHIP_OK = 0
BHIP_ERR_INVALID = -1
BHIP_ERR_UNSUPPORTED = -2
BHIP_ERR_HIP = -3
BHIP_ERR_NOMEM = -4
BHIP_ERR_CAPACITY = -5


class FhCfg(C.Structure):
    _fields_ = [("detectThreshold", C.c_float), ("extractRadius", C.c_int), ("maxFeaturesPerScale", C.c_int), ("initialSampleSize", C.c_int),
                ("initialSize", C.c_int), ("numberScalesPerOctave", C.c_int), ("numberOfOctaves", C.c_int), ("scaleStepSize", C.c_int)]


class SurfCfg(C.Structure):
    _fields_ = [("widthLargeGrid", C.c_int), ("widthSubRegion", C.c_int), ("widthSample", C.c_int), ("weightSigma", C.c_double),
                ("overLap", C.c_int), ("sigmaLargeGrid", C.c_double), ("sigmaSubRegion", C.c_double)]


class OriCfg(C.Structure):
    _fields_ = [("objectRadiusToScale", C.c_double), ("samplePeriod", C.c_double), ("windowSize", C.c_double), ("radius", C.c_int),
                ("weightSigma", C.c_double), ("sampleWidth", C.c_int)]


P = C.POINTER
_vp, _i, _f, _d, _ll = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong
_fp, _dp, _ip, _u8p, _i16p, _i32p, _llp = P(C.c_float), P(C.c_double), P(C.c_int), P(C.c_uint8), P(C.c_int16), P(C.c_int32), P(C.c_longlong)

# name -> (restype, argtypes); must list every symbol of include/boofhip.h (tests/test_cabi_symbols.py checks both directions)
SIGNATURES = {
    "bhip_fh_cfg_default": (None, [P(FhCfg)]),
    "bhip_surf_cfg_default": (None, [P(SurfCfg)]),
    "bhip_ori_cfg_default": (None, [P(OriCfg), _i]),
    "bhip_ctx_create": (_i, [_i, P(_vp)]),
    "bhip_ctx_create_on_stream": (_i, [_i, _vp, P(_vp)]),
    "bhip_ctx_destroy": (_i, [_vp]),
    "bhip_ctx_synchronize": (_i, [_vp]),
    "bhip_last_error": (C.c_char_p, [_vp]),
    "bhip_host_alloc": (_i, [_vp, C.c_longlong, P(_vp)]),
    "bhip_host_free": (_i, [_vp]),
    "bhip_version": (C.c_char_p, []),
    "bhip_profile_enable": (_i, [_vp, _i]),
    "bhip_profile_reset": (_i, [_vp]),
    "bhip_profile_report": (_i, [_vp, C.c_char_p, _i]),
    "bhip_surf_create": (_i, [_vp, P(FhCfg), P(SurfCfg), P(OriCfg), _i, P(_vp)]),
    "bhip_surf_destroy": (_i, [_vp]),
    "bhip_surf_detect_f32": (_i, [_vp, P(_fp), _ip, _ip, _i, _i, _i]),
    "bhip_surf_detect_dev_f32": (_i, [_vp, _vp, _ll, _i, _i, _i, _i]),
    "bhip_surf_detect_u8": (_i, [_vp, P(_u8p), _ip, _ip, _i, _i, _i]),
    "bhip_surf_detect_planar_f32": (_i, [_vp, P(_fp), _i, _i, _i, _i, _i]),
    "bhip_surf_count": (_i, [_vp, _i, _ip]),
    "bhip_surf_counts": (_i, [_vp, _ip, _i]),
    "bhip_surf_fetch": (_i, [_vp, _i, _dp, _dp, _u8p, _dp]),
    "bhip_surf_fetch_all": (_i, [_vp, _dp, _dp, _u8p, _dp]),
    "bhip_assoc_l2_surf": (_i, [_vp, _i, _ip, _ip, _d, _i, _ip, _dp]),
    "bhip_surf_create_brief": (_i, [_vp, P(FhCfg), _i, _i, _i32p, _i32p, P(_vp)]),
    "bhip_surf_fetch_brief": (_i, [_vp, _i, _i32p]),
    "bhip_surf_dev_view_brief": (_i, [_vp, _i, P(_vp), _ip, _ip]),
    "bhip_assoc_hamming_surf": (_i, [_vp, _i, _ip, _ip, _d, _i, _ip, _dp]),
    "bhip_surf_dev_view": (_i, [_vp, _i, P(_vp), P(_vp), P(_vp), _ip]),
    "bhip_surf_dof": (_i, [_vp]),
    "bhip_surf_total": (_i, [_vp, _llp]),
    "bhip_surf_describe_points": (_i, [_vp, _i, _dp, _i, _dp, _u8p, _dp]),
    "bhip_surf_fetch_integral": (_i, [_vp, _i, _fp]),
    "bhip_integral_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_hessian_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_nonmax_block_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _f, _i, _i16p, _i, _ip]),
    "bhip_select_nbest_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i16p, _i, _i, _i, _i16p, _ip]),
    "bhip_fh_detect_f32": (_i, [_vp, P(FhCfg), _fp, _i, _i, _i, _i, _dp, _i, _ip]),
    "bhip_assoc_l2_f64": (_i, [_vp, _dp, _i, _dp, _i, _i, _d, _i, _i, _ip, _dp]),
    "bhip_assoc_hamming": (_i, [_vp, _i32p, _i, _i32p, _i, _i, _d, _i, _ip, _dp]),
    "bhip_assoc_l2_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _d, _i, _i, _vp, _vp]),
    "bhip_assoc_hamming_dev": (_i, [_vp, _vp, _i, _vp, _i, _i, _d, _i, _vp, _vp]),
    "bhip_assoc_l2_dev_batched": (_i, [_vp, _vp, _vp, _i, _i, _llp, _ip, _llp, _ip, _d, _i, _vp, _vp]),
    "bhip_assoc_hamming_dev_batched": (_i, [_vp, _vp, _vp, _i, _i, _llp, _ip, _llp, _ip, _d, _i, _vp, _vp]),
    "bhip_assoc_l2_shard_phase1": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _d, _vp, _vp, _vp]),
    "bhip_assoc_hamming_shard_phase1": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _d, _vp, _vp, _vp]),
    "bhip_assoc_shard_phase2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "bhip_assoc_coltop_bytes": (_i, []),
    "bhip_conv_h_f32": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_conv_v_f32": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_conv_norm_h_f32": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_conv_norm_v_f32": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_gaussian_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _d, _i, _fp, _i, _i]),
    "bhip_sobel_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _fp, _i, _i, _i]),
    "bhip_three_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _fp, _i, _i, _i]),
    "bhip_conv_down_norm_h_f32": (_i, [_vp, _fp, _i, _fp, _i, _i, _i, _i, _fp, _i, _i, _i, _i, _i]),
    "bhip_conv_down_norm_v_f32": (_i, [_vp, _fp, _i, _fp, _i, _i, _i, _i, _fp, _i, _i, _i, _i, _i]),
    "bhip_gaussian_kernel1d_f32": (_i, [_d, _i, _fp, _i]),
    "bhip_pyramid_layout": (_i, [_i, _i, _ip, _i, _ip, _llp, _llp]),
    "bhip_pyramid_f32": (_i, [_vp, _fp, _i, _ip, _i, _fp, _i, _i, _i, _i, _fp]),
    "bhip_pyramid_dev_f32": (_i, [_vp, _fp, _i, _ip, _i, _vp, _ll, _i, _i, _i, _i, _vp]),
    "bhip_conv2d_f32": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_mean_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_median_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_corner_intensity_f32": (_i, [_vp, _i, _i, _f, _fp, _fp, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_integral_u8_s32": (_i, [_vp, _u8p, _i, _i, _i, _i, _i32p, _i, _i]),
    "bhip_hessian_s32": (_i, [_vp, _i32p, _i, _i, _i, _i, _i, _i, _fp, _i, _i]),
    "bhip_fh_detect_s32": (_i, [_vp, P(FhCfg), _i32p, _i, _i, _i, _i, _dp, _i, _ip]),
    "bhip_brief_u8": (_i, [_vp, _u8p, _i, _i, _i, _i, _i, _i, _i32p, _i32p, _dp, _i, _i32p]),
    "bhip_brief_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _i32p, _i32p, _dp, _i, _i32p]),
    "bhip_conv_h_dev_f32": (_i, [_vp, _fp, _i, _i, _vp, _ll, _i, _i, _i, _i, _vp, _ll, _i]),
    "bhip_conv_v_dev_f32": (_i, [_vp, _fp, _i, _i, _vp, _ll, _i, _i, _i, _i, _vp, _ll, _i]),
    "bhip_conv_norm_h_dev_f32": (_i, [_vp, _fp, _i, _i, _vp, _ll, _i, _i, _i, _i, _vp, _ll, _i]),
    "bhip_conv_norm_v_dev_f32": (_i, [_vp, _fp, _i, _i, _vp, _ll, _i, _i, _i, _i, _vp, _ll, _i]),
    "bhip_gaussian_dev_f32": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _d, _i, _vp, _ll, _i]),
    "bhip_sobel_dev_f32": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _vp, _vp, _ll, _i, _i]),
    "bhip_three_dev_f32": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _vp, _vp, _ll, _i, _i]),
    "bhip_gradient_intensity_dev_f32": (_i, [_vp, _i, _vp, _vp, _ll, _i, _i, _i, _i, _vp, _ll, _i]),
    "bhip_nonmax_block_dev_f32": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "bhip_corner_intensity_dev_f32": (_i, [_vp, _i, _i, _f, _vp, _vp, _ll, _i, _i, _i, _i, _vp, _ll, _i]),
    "bhip_brief_dev_f32": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i32p, _i32p, _vp, _ip, _vp]),
}

_lib = None


class BoofHipMissing(RuntimeError):
    pass


def _preload_process_hip_runtime():
    """One process must use ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's);
    if libboofhip.so pulled in the system copy first and torch were imported later, the process would hold two runtimes and the
    second one finds no GPU.  So when a torch wheel with a bundled runtime is installed, load that copy first (without importing
    torch); libboofhip.so's NEEDED libamdhip64.so.7 then binds to it and device pointers / streams can be shared with torch."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libboofhip.so (no GPU needed for this step) and attach the signatures."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BoofHipMissing("%s not found: build it with `python -m boofcv_amd.build` (hipcc, gfx950). "
                                 "There is no CPU fallback." % LIB_PATH)
        _preload_process_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here means the library and the header disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
