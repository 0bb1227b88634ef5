"""CPU-side checks of the drop-in boundary: libboofhip.so loads without a GPU and exports exactly what include/boofhip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "boofhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bhip_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from boofcv_amd import build, _lib
    build.build()
    L = _lib.load()
    declared = _declared()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), "libboofhip.so does not export " + name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert b"gfx950" in L.bhip_version()


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the context constructor must fail loudly (no silent CPU path)."""
    import ctypes as C
    from boofcv_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    st = L.bhip_ctx_create(0, C.byref(h))
    if st == 0:  # running on a GPU box
        L.bhip_ctx_destroy(h)
        pytest.skip("a GPU is present")
    assert st < 0 and not h.value


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "boofcv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle|liboracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, "product files reference the oracle: %s" % bad


def test_config_defaults_match_reference():
    from boofcv_amd import _lib
    L = _lib.load()
    fh = _lib.FhCfg(); L.bhip_fh_cfg_default(fh)
    assert (fh.detectThreshold, fh.extractRadius, fh.maxFeaturesPerScale, fh.initialSampleSize, fh.initialSize, fh.numberScalesPerOctave,
            fh.numberOfOctaves, fh.scaleStepSize) == (1.0, 2, -1, 1, 9, 4, 4, 6)  # ConfigFastHessian.java:33-70
    sd = _lib.SurfCfg(); L.bhip_surf_cfg_default(sd)
    assert (sd.widthLargeGrid, sd.widthSubRegion, sd.widthSample, sd.weightSigma, sd.overLap, sd.sigmaLargeGrid, sd.sigmaSubRegion) == (4, 5, 3, 4.5, 2, 2.5, 2.5)
    o = _lib.OriCfg(); L.bhip_ori_cfg_default(o, 1)
    assert (o.objectRadiusToScale, o.samplePeriod, o.radius, o.weightSigma, o.sampleWidth) == (0.5, 0.65, 8, -1.0, 6) and abs(o.windowSize - 3.141592653589793 / 3) < 1e-15
    L.bhip_ori_cfg_default(o, 0)
    assert (o.samplePeriod, o.radius, o.sampleWidth) == (1.0, 6, 6)


def test_shipped_library_has_no_experiment_switches():
    """Ablation / tile-variant / stamp switches exist only in the -DBHIP_EXPERIMENTS build (libboofhip_exp.so): an environment variable must
    never be able to make a production kernel skip work.  The parity cross-check hooks (UNFUSED, NOSHARE, EXACT, ...) stay."""
    from boofcv_amd import build
    blob = open(build.LIB, "rb").read()
    for name in (b"BHIP_FUSED_ABLATE", b"BHIP_FUSED_VARIANT", b"BHIP_DESCRIBE_LDSPAD", b"BHIP_DESCRIBE_STAMPS", b"BHIP_DESCRIBE_NOORDER", b"BHIP_ASSOC_ABLATE"):
        assert name not in blob, name
    assert b"BHIP_DETECT_UNFUSED" in blob


def test_image_views_are_validated_before_raw_pointers_leave_python():
    """api.GrayF32 / GrayU8 reject views that do not fit their array (the C side only sees a pointer)."""
    import numpy as np
    from boofcv_amd import api
    data = np.zeros(100, np.float32)
    api.GrayF32(10, 10, data)
    for bad in (dict(width=10, height=11), dict(width=10, height=10, startIndex=1), dict(width=10, height=10, stride=11), dict(width=10, height=10, stride=9)):
        kw = dict(width=10, height=10, startIndex=0, stride=None)
        kw.update(bad)
        with pytest.raises(api.IllegalArgumentException):
            api.GrayF32(kw["width"], kw["height"], data, kw["startIndex"], kw["stride"])
    img = api.GrayF32(10, 10, data)
    img.subimage(2, 3, 10, 10)
    for box in ((2, 3, 11, 10), (-1, 0, 5, 5), (6, 0, 5, 5)):
        with pytest.raises(api.IllegalArgumentException):
            img.subimage(*box)
    with pytest.raises(api.IllegalArgumentException):
        api.GrayU8(4, 4, np.zeros(15, np.uint8))
