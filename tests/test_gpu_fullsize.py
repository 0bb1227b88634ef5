"""GPU parity at the sizes BASELINE.json quotes (configs 2-5): direct comparison with the CPU oracle where it finishes in seconds, plus the
size-independent properties of the domain (batch == single, planted matches recovered, sub-sampled layer == blurred image).
`pytest -m gpu`; every call goes through the C ABI."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def api():
    from boofcv_amd import api as a
    a.Context.default()
    return a


def G(api, g):
    return api.GrayF32(g.width, g.height, g.buf, g.startIndex, g.stride)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def blobs(orc, w, h, seed):
    """S-blobs built with numpy: background 50 + Gaussian blobs + U[0,2) noise."""
    rng = np.random.default_rng(seed)
    n = max(1, w * h // 2000)
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    sig = rng.choice([2.0, 3.0, 5.0, 8.0, 13.0, 21.0], n)
    amp = rng.uniform(40, 100, n) * rng.choice([-1.0, 1.0], n)
    xs, ys = np.arange(w, dtype=np.float64)[None, :], np.arange(h, dtype=np.float64)[:, None]
    gx = np.exp(-((xs - cx[:, None]) ** 2) / (2 * sig[:, None] ** 2))
    gy = np.exp(-((ys - cy[None, :]) ** 2) / (2 * sig[None, :] ** 2)) * amp[None, :]
    return orc.Gray.from_array((50.0 + gy @ gx + rng.uniform(0, 2, (h, w))).astype(np.float32))


# ------------------------------------------------------------------------------------------------------------------ config 2
def test_config2_1080p_detect_describe(api, orc):
    """Two 1920x1080 frames (one S-blobs, one dense S-noise) as a batch: key points bit-exact and in the reference's order,
    descriptors inside the 1e-5 bar, batch == frame-by-frame."""
    frames = [blobs(orc, 1920, 1080, 1000), orc.noise_image(1920, 1080, 234)]
    dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32)
    dd.detectBatch([G(api, f) for f in frames])
    batch = [tuple(np.array(x) for x in dd._results(i)) for i in range(2)]
    ref = orc.Surf(True)
    for i, f in enumerate(frames):
        ref.detect(f, threads=THREADS)
        pts, ang, white, desc = ref.fetch()
        got = batch[i]
        assert len(pts) > (1500 if i == 0 else 20000)
        assert np.array_equal(got[0], pts) and np.array_equal(got[2], white)
        derr = np.max(np.abs(got[3] - desc), axis=1)
        assert int((derr > 1e-5).sum()) == 0, (i, int((derr > 1e-5).sum()), derr.max())     # every descriptor, no exception list
        assert np.abs(np.angle(np.exp(1j * (got[1] - ang)))).max() < 1e-12            # every orientation
        dd.detect(G(api, f))
        single = dd._results()
        assert all(np.array_equal(a, b) for a, b in zip(single, got))
        # the integral image kept for the describe stage is the reference's, bit for bit
        assert np.array_equal(bits(dd.fetchIntegral(0, 1920, 1080)), bits(orc.integral(f).array()))


# ------------------------------------------------------------------------------------------------------------------ config 4
def test_config4_brief512_16384_hamming(api, orc):
    """16384 x 16384 BRIEF-512 greedy Hamming association with mutual-best validation: bit-exact against the oracle, the planted
    matches are recovered, and the 8-way row-sharded form (phase 1 per slice, records concatenated as the all-gather would,
    phase 2) reproduces the unsharded answer."""
    n, words = 16384, 16
    rng = np.random.default_rng(4)
    a = rng.integers(-2 ** 31, 2 ** 31, (n, words), dtype=np.int64).astype(np.int32)
    b = a.copy()
    k = 3 * n // 4
    for row in range(k):   # up to 64 random bit flips
        nf = rng.integers(0, 65)
        pos = rng.choice(512, nf, replace=False)
        np.bitwise_xor.at(b[row].view(np.uint32), pos // 32, (np.uint32(1) << (pos % 32).astype(np.uint32)))
    b[k:] = rng.integers(-2 ** 31, 2 ** 31, (n - k, words), dtype=np.int64).astype(np.int32)
    perm = rng.permutation(n)
    b = np.ascontiguousarray(b[perm])
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    assoc = api.FactoryAssociation.greedy(api.ScoreAssociateHamming_B(), api.Double_MAX_VALUE, True)
    assoc.setSource(a); assoc.setDestination(b); assoc.associate()
    pairs, fit = assoc.getPairs(), assoc.getFitQuality()
    ep, ef = orc.associate_hamming(a, b, orc.MAX_VALUE_F64, True, threads=THREADS)
    assert np.array_equal(pairs, ep) and np.array_equal(fit, ef)
    # planted pairs: at most 64 of 512 bits differ, random pairs differ in ~256 +- 11
    planted = np.arange(k)
    assert (pairs[planted] == inv[planted]).mean() > 0.999
    assert fit[planted].max() <= 64
    # sharded 8-way on one GPU
    import torch
    from boofcv_amd import sharded
    eng = sharded.GpuEngine(device=0)
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    part = sharded.row_partition(n, 8)
    outs = [eng.phase1("hamming", d_a[lo:lo + cnt].contiguous(), lo, d_b, api.Double_MAX_VALUE) for lo, cnt in part]
    allrec = torch.cat([o[2] for o in outs])
    got = [eng.phase2(allrec, 8, n, o[0], o[1], lo) for (lo, cnt), o in zip(part, outs)]
    torch.cuda.synchronize()
    assert np.array_equal(np.concatenate([g[0].cpu().numpy() for g in got]), ep)
    assert np.array_equal(np.concatenate([g[1].cpu().numpy() for g in got]), ef)


# ------------------------------------------------------------------------------------------------------------------ config 5
def test_config5_4k_pyramid_gradient_chain(api, orc):
    """3840x2160: pyramid [1,2,4,8] (Gaussian radius 2) bit-exact, Sobel of every layer bit-exact, and the layer property of
    TestPyramidDiscreteSampleBlur (layer i == blurred layer i-1 at every second pixel)."""
    img = blobs(orc, 3840, 2160, 5000)
    scales = [1, 2, 4, 8]
    pyr = api.FactoryPyramid.discreteGaussian(scales, -1, 2).process(G(api, img))
    ker = orc.gaussian1d_f32(-1, 2)
    exp_layers, _ = orc.pyramid(ker, -1, scales, img)
    for i, e in enumerate(exp_layers):
        layer = pyr.getLayer(i)
        assert np.array_equal(bits(layer.array()), bits(e)), i
        dx, dy = api.GrayF32(layer.width, layer.height), api.GrayF32(layer.width, layer.height)
        api.GradientSobel.process(layer, dx, dy, 0)
        ex, ey = orc.gradient("sobel", orc.Gray.from_array(e), border_zero=True)
        assert np.array_equal(bits(dx.array()), bits(ex.array())) and np.array_equal(bits(dy.array()), bits(ey.array())), i
        if i > 0:
            prev = pyr.getLayer(i - 1)
            blurred = api.BlurImageOps.gaussian(prev, None, -1, 2).array()
            assert np.abs(blurred[::2, ::2][:layer.height, :layer.width] - layer.array()).max() < 1e-3


def test_config5_4k_device_chain(api, orc):
    """BASELINE config 5 end to end on a device-resident 3840x2160 frame, without leaving HBM between the stages:
    pyramid [1,2,4,8] (Gaussian r=2, PyramidDiscreteSampleBlur.java:88-118) -> Sobel of every layer (GradientSobel.java:158-173) -> |grad|^2 ->
    strict block NMS r=2 (NonMaxBlock.java:69-94) -> Fast-Hessian + SURF-64 (stable) on layer 0.  Every stage against the oracle:
    images and NMS lists bit-exact (lists in the reference's order), key points bit-exact, descriptors within 1e-5."""
    torch = pytest.importorskip("torch")
    from boofcv_amd import device as dv
    ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
    ops = dv.DeviceImageOps(ctx)
    W, H = 3840, 2160
    img = blobs(orc, W, H, 5000)
    frames = torch.from_numpy(img.array().copy()).cuda().unsqueeze(0)
    scales = [1, 2, 4, 8]
    ker = orc.gaussian1d_f32(-1, 2)
    layers = ops.pyramid(ker, scales, frames)
    exp_layers, _ = orc.pyramid(ker, -1, scales, img)
    total_nms = 0
    for i, e in enumerate(exp_layers):
        assert np.array_equal(bits(layers[i][0].cpu().numpy()), bits(e)), i
        dx, dy = ops.sobel(layers[i], 0)
        sq = ops.intensity(dv.INTENSITY_SQ, dx, dy)
        xy, n = ops.nonmax(sq, 2, 25.0, 2)
        ctx.synchronize()
        ex, ey = orc.gradient("sobel", orc.Gray.from_array(e), border_zero=True)
        assert np.array_equal(bits(dx[0].cpu().numpy()), bits(ex.array())) and np.array_equal(bits(dy[0].cpu().numpy()), bits(ey.array())), i
        esq = ex.array() * ex.array() + ey.array() * ey.array()
        assert np.array_equal(bits(sq[0].cpu().numpy()), bits(esq)), i
        elist = orc.nonmax(orc.Gray.from_array(esq), 2, 25.0, 2, threads=THREADS)
        cnt = int(n[0].item())
        assert cnt == len(elist) and np.array_equal(xy[0, :cnt].cpu().numpy(), elist), (i, cnt, len(elist))
        total_nms += cnt
    assert total_nms > 1000
    # FH + SURF on layer 0 (== the input frame), device resident
    dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
    l0 = layers[0]
    dd.detectDevice(l0.data_ptr(), l0.stride(0), l0.stride(1), W, H, 1)
    ref = orc.Surf(True)
    n = ref.detect(img, threads=THREADS)
    xys, ang, white, desc = ref.fetch()
    got = dd._results(0)
    assert dd.getNumberOfFeatures() == n and n > 3000
    assert np.array_equal(got[0], xys) and np.array_equal(got[2], white)
    derr = np.max(np.abs(got[3] - desc), axis=1)
    assert int((derr > 1e-5).sum()) == 0, "descriptors outside 1e-5: %d, max %.3g" % (int((derr > 1e-5).sum()), derr.max())
    dang = np.abs(np.angle(np.exp(1j * (got[1] - ang))))
    assert dang.max() < 1e-9, dang.max()


# ------------------------------------------------------------------------------------------------------------------ config 2 at its real batch
@pytest.mark.parametrize("B,route", [(256, "device"), (130, "host")])
def test_config2_full_batch_every_frame(api, orc, B, route):
    """BASELINE config 2 at batch size: 256 device-resident 1920x1080 S-blobs frames through bhip_surf_detect_dev_f32 (the bench path: fused
    integral kernel, capacity regrow, tile-ordered / XCD-chunked describe over ~550 k key points) and 130 host frames through
    bhip_surf_detect_f32.  EVERY frame's key points are compared with the oracle bit for bit (count, order, location, scale, Laplacian
    sign); descriptors and orientations of every 8th frame against the 1e-5 / 1e-12 bars with no exception list."""
    torch = pytest.importorskip("torch")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    W, H = 1920, 1080
    frames = bench.synth_frames(B, H, W, 1000, torch.device("cuda", 0))
    torch.cuda.synchronize()
    host = frames.cpu().numpy()
    if route == "device":
        ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
        dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
        dd.detectDevice(frames.data_ptr(), H * W, W, W, H, B)
    else:
        del frames
        dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32)
        dd.detectBatch([api.GrayF32.wrap(host[i]) for i in range(B)])
    ref = orc.Surf(True)
    total = 0
    for i in range(B):
        n = ref.detect(orc.Gray.from_array(host[i]), threads=THREADS)
        xys, ang, white, desc = ref.fetch()
        got = dd._results(i)
        assert len(got[0]) == n and np.array_equal(got[0], xys) and np.array_equal(got[2], white), i
        if i % 8 == 0:
            derr = np.max(np.abs(got[3] - desc), axis=1)
            assert int((derr > 1e-5).sum()) == 0, (i, int((derr > 1e-5).sum()), derr.max())
            assert np.abs(np.angle(np.exp(1j * (got[1] - ang)))).max() < 1e-12, i
        total += n
    assert total == dd.totalFeatures() and total > 1500 * B
