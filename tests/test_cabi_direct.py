"""The C ABI called from C: tests/cabi_direct.c is compiled with plain gcc against libboofhip.so and run as its own process (no ctypes,
no Python objects on the path); its dumped results must equal the oracle's -- and therefore what the ctypes route returns."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi_direct.c")


def _build(tmp):
    from boofcv_amd import build
    lib = build.build()
    exe = os.path.join(tmp, "cabi_direct")
    libdir = os.path.dirname(lib)
    hipdirs = ["/opt/rocm/lib"]
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            hipdirs.insert(0, os.path.join(os.path.dirname(spec.origin), "lib"))
    except Exception:
        pass
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", libdir, "-lboofhip",
           "-Wl,-rpath," + libdir] + ["-Wl,-rpath-link," + d for d in hipdirs] + ["-Wl,-rpath," + d for d in hipdirs]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cabi_direct_compiles_and_links(tmp_path):
    """CPU: plain C sees the header and the library's exports (no GPU call)."""
    exe = _build(str(tmp_path))
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 1 and "usage:" in p.stderr


@pytest.mark.gpu
def test_cabi_direct_matches_oracle(tmp_path, orc):
    exe = _build(str(tmp_path))
    W, H = 320, 240
    imgs = [orc.noise_image(W, H, 234), orc.noise_image(W, H, 235)]
    paths = []
    for i, im in enumerate(imgs):
        p = str(tmp_path / ("img%d.f32" % i))
        im.array().astype(np.float32).tofile(p)
        paths.append(p)
    out = str(tmp_path / "out.bin")
    p = subprocess.run([exe, str(W), str(H), paths[0], paths[1], out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    raw = open(out, "rb").read()
    n = np.frombuffer(raw, np.int32, 2)
    off = 8
    got = []
    for k in range(2):
        m = int(n[k])
        xys = np.frombuffer(raw, np.float64, 3 * m, off).reshape(m, 3); off += 24 * m
        ang = np.frombuffer(raw, np.float64, m, off); off += 8 * m
        white = np.frombuffer(raw, np.uint8, m, off); off += m
        desc = np.frombuffer(raw, np.float64, 64 * m, off).reshape(m, 64); off += 512 * m
        got.append((xys, ang, white, desc))
    pairs = np.frombuffer(raw, np.int32, int(n[0]), off); off += 4 * int(n[0])
    fit = np.frombuffer(raw, np.float64, int(n[0]), off)
    ref = orc.Surf(True)
    for k in range(2):
        assert ref.detect(imgs[k]) == n[k] and n[k] > 100
        xys, ang, white, desc = ref.fetch()
        assert np.array_equal(got[k][0], xys) and np.array_equal(got[k][2], white)
        assert np.abs(got[k][3] - desc).max() < 1e-5
    ep, ef = orc.associate_l2(got[0][3], got[1][3])
    assert np.array_equal(pairs, ep) and np.array_equal(fit, ef)
