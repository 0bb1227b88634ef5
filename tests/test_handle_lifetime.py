"""Handle lifetime rules of the C ABI (include/boofhip.h, "handles may be destroyed in any order and more than once").

Round 2's GPU logs showed `std::bad_variant_access` + a core dump (rc 134) at interpreter exit whenever a GPU test had failed: a failed
test keeps ctx and the detect+describe object alive in its traceback, the garbage collector then finalised them in arbitrary order, and
bhip_surf_destroy read the stream of an already deleted bhip_ctx and handed that garbage to the HIP runtime.  The library now keeps a
registry of live handles; these tests pin the behaviour, in child processes so that a crash is an observable exit code.
"""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, timeout=300):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_destroy_refuses_pointers_that_are_not_live_handles():
    """No GPU needed: a pointer the registry does not know is refused, never dereferenced (it points at zeroed host memory here,
    the old code would have read a device index and a stream out of it and called the HIP runtime with them)."""
    from boofcv_amd import _lib
    L = _lib.load()
    junk = C.create_string_buffer(4096)
    p = C.c_void_p(C.addressof(junk))
    assert L.bhip_ctx_destroy(p) == _lib.BHIP_ERR_INVALID
    assert L.bhip_surf_destroy(p) == _lib.BHIP_ERR_INVALID
    assert L.bhip_ctx_destroy(None) == _lib.BHIP_OK and L.bhip_surf_destroy(None) == _lib.BHIP_OK
    h = C.c_void_p()
    assert L.bhip_surf_create(p, None, None, None, 1, C.byref(h)) == _lib.BHIP_ERR_INVALID and not h.value


@pytest.mark.gpu
def test_context_closed_before_its_detector():
    """`ctx.close(); del dd` (and the same order through the raw C ABI) must neither crash nor leak into the HIP runtime."""
    r = _child("""
import ctypes as C, numpy as np
from boofcv_amd import api, _lib
L = _lib.load()
ctx = api.Context(0)
dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
img = api.GrayF32(160, 120, (np.random.default_rng(1).random(160 * 120) * 100).astype(np.float32))
dd.detect(img)
n = dd.getNumberOfFeatures()
raw = dd._h
ctx.close()                                   # closes dd first (Python side)
assert dd._h is None and ctx._h is None
assert L.bhip_surf_destroy(raw) == _lib.BHIP_ERR_INVALID      # already destroyed: refused, not dereferenced
del dd
# raw C ABI, wrong order, no Python bookkeeping
c = C.c_void_p(); s = C.c_void_p()
assert L.bhip_ctx_create(0, C.byref(c)) == 0
assert L.bhip_surf_create(c, None, None, None, 1, C.byref(s)) == 0
ptr = (C.POINTER(C.c_float) * 1)(img.data.ctypes.data_as(C.POINTER(C.c_float)))
z = (C.c_int * 1)(0); st = (C.c_int * 1)(160)
assert L.bhip_surf_detect_f32(s, ptr, z, st, 160, 120, 1) == 0
assert L.bhip_ctx_destroy(c) == 0              # context first: the detector becomes an inert shell
cnt = C.c_int(-1)
assert L.bhip_surf_count(s, 0, C.byref(cnt)) == _lib.BHIP_ERR_INVALID
assert L.bhip_surf_detect_f32(s, ptr, z, st, 160, 120, 1) == _lib.BHIP_ERR_INVALID
assert L.bhip_surf_destroy(s) == 0
assert L.bhip_surf_destroy(s) == _lib.BHIP_ERR_INVALID
assert L.bhip_ctx_destroy(c) == _lib.BHIP_ERR_INVALID
print("ok", n)
""")
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("ok")


@pytest.mark.gpu
def test_failing_process_exits_with_its_own_code():
    """What a failed GPU test looks like to the interpreter: live ctx + detector + a ctx on torch's stream, kept alive by a traceback
    (a reference cycle), then an uncaught exception.  Must be rc 1 -- round 2 got rc 134 (abort) here."""
    r = _child("""
import sys, numpy as np, torch
from boofcv_amd import api
keep = []
def work():
    ctx = api.Context(0)
    tctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
    dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
    dd2 = api.FactoryDetectDescribe.surfFast(None, None, None, api.GrayF32, ctx=tctx)
    frames = torch.rand((2, 120, 160), device="cuda:0") * 100
    dd2.detectDevice(frames.data_ptr(), 120 * 160, 160, 160, 120, 2)
    img = api.GrayF32(160, 120, (np.random.default_rng(2).random(160 * 120) * 100).astype(np.float32))
    dd.detect(img)
    try:
        raise AssertionError("descriptors outside 1e-5")
    except AssertionError as e:
        keep.append(e)           # traceback -> frame -> ctx, tctx, dd, dd2, frames: finalised at interpreter exit, in no particular order
        cyc = [e]; cyc.append(cyc)
        raise
work()
""")
    assert r.returncode == 1, "rc %d\n%s\n%s" % (r.returncode, r.stdout, r.stderr)
    assert "AssertionError" in r.stderr and "terminate called" not in r.stderr and "core dumped" not in r.stderr
