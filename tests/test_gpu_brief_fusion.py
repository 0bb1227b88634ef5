"""Fast-Hessian + BRIEF as one detect+describe object, resident on the device, feeding the Hamming association
(FactoryDetectDescribe.fuseTogether(fastHessian, null, brief): F:factory/feature/detdesc/FactoryDetectDescribe.java:279-284,
F:abst/feature/detdesc/DetectDescribeFusion.java:95-127, F:abst/feature/describe/WrapDescribeBrief.java:47-58,
F:alg/feature/describe/DescribePointBrief.java:71-89) against the oracle's separate stages: points bit-exact and in the reference's order,
words bit-exact including border points, association pairs and scores identical.  `pytest -m gpu`; every call goes through the C ABI."""
import os

import numpy as np
import pytest

THREADS = min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def api():
    from boofcv_amd import api as a
    a.Context.default()
    return a


def G(api, g):
    return api.GrayF32(g.width, g.height, g.buf, g.startIndex, g.stride)


def blobs(orc, w, h, seed):
    rng = np.random.default_rng(seed)
    n = max(1, w * h // 2000)
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    sig = rng.choice([2.0, 3.0, 5.0, 8.0, 13.0, 21.0], n)
    amp = rng.uniform(40, 100, n) * rng.choice([-1.0, 1.0], n)
    xs, ys = np.arange(w, dtype=np.float64)[None, :], np.arange(h, dtype=np.float64)[:, None]
    gx = np.exp(-((xs - cx[:, None]) ** 2) / (2 * sig[:, None] ** 2))
    gy = np.exp(-((ys - cy[None, :]) ** 2) / (2 * sig[None, :] ** 2)) * amp[None, :]
    return orc.Gray.from_array((50.0 + gy @ gx + rng.uniform(0, 2, (h, w))).astype(np.float32))


def fuse(api, cfg=None, definition=None, imageType=None, ctx=None):
    brief = api.FactoryDescribeRegionPoint.brief(None, imageType or api.GrayF32, definition=definition, ctx=ctx)
    return api.FactoryDetectDescribe.fuseTogether(api.FactoryInterestPoint.fastHessian(cfg), None, brief, ctx=ctx)


def reference(orc, img, sp, cp):
    """the reference's stages one after the other on the CPU: integral -> Fast-Hessian -> BRIEF at every point, on the unblurred frame"""
    pts = orc.fh_detect(orc.integral(img), threads=THREADS)
    return pts, orc.brief_describe(img, pts[:, :2].copy(), 16, sp, cp)


def test_definition_and_argument_checks_cpu():
    """(no GPU) the product's FactoryBriefDefinition.gaussian2 / java.util.Random restatement gives the oracle's table; what the GPU
    fusion does not cover is refused the way an override hook refuses it (RuntimeError -> "use the Java path")"""
    from boofcv_amd import api
    from oracle import pyoracle as orc
    orc.build()
    d = api.FactoryBriefDefinition.gaussian2(api.Random(123), 16, 512)
    sp, cp = orc.brief_definition()
    assert np.array_equal(d.samplePoints, sp) and np.array_equal(d.compare, cp) and d.getLength() == 512
    assert api.Random(0).nextInt() == -1155484576 and api.Random(0).nextDouble() == 0.730967787376657
    with pytest.raises(RuntimeError):
        api.FactoryDescribeRegionPoint.brief(api.ConfigBrief(fixed=False), api.GrayF32)
    class Other: pass
    with pytest.raises(RuntimeError):
        api.FactoryDetectDescribe.fuseTogether(Other(), None, Other())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kind", [(640, 480, "noise"), (331, 250, "noise"), (1920, 1080, "blobs")])
def test_fast_hessian_brief_fusion_parity(api, orc, w, h, kind):
    sp, cp = orc.brief_definition()
    img = orc.noise_image(w, h, 234) if kind == "noise" else blobs(orc, w, h, 1000)
    dd = fuse(api)
    dd.detect(G(api, img))
    pts, words = reference(orc, img, sp, cp)
    got = dd._results(0)
    assert dd.getNumberOfFeatures() == len(pts) and len(pts) > 200
    assert np.array_equal(got[0], pts), "points differ from the reference order / values"
    assert np.array_equal(got[3], words), "BRIEF words differ"
    # border points exist in this set (the F32 border rule: a pair outside the frame is skipped without shifting) and are covered
    c = pts[:, :2].astype(np.int64)
    border = (c[:, 0] - 16 < 0) | (c[:, 0] + 16 >= w) | (c[:, 1] - 16 < 0) | (c[:, 1] + 16 >= h)
    if kind == "noise":
        assert border.any()
    # interface facts of DetectDescribeFusion / WrapFHtoInterestPoint
    assert dd.getOrientation(0) == 0.0 and not dd.hasOrientation() and dd.hasScale()
    assert dd.getRadius(3) == pts[3, 2] * 2.0
    f = dd.getDescription(5)
    assert f.numBits == 512 and np.array_equal(f.data, words[5])
    assert dd.getDescriptionType() is api.TupleDesc_B and dd.createDescription().data.shape == (16,)


@pytest.mark.gpu
def test_fusion_batch_device_and_resident_hamming_association(api, orc):
    """three frames as one batch (host frames, then the same frames device-resident): every image equals its single-frame result, and
    the Hamming association on the resident words equals the oracle's greedy association of the fetched words -- pairs and scores"""
    import torch
    sp, cp = orc.brief_definition()
    imgs = [orc.noise_image(400, 300, 234 + i) for i in range(3)]
    dd = fuse(api)
    dd.detectBatch([G(api, g) for g in imgs])
    refs = [reference(orc, g, sp, cp) for g in imgs]
    for i, (pts, words) in enumerate(refs):
        r = dd._results(i)
        assert np.array_equal(r[0], pts) and np.array_equal(r[3], words)
    xys, words_all, starts = dd.fetchAll()
    assert np.array_equal(words_all, np.concatenate([r[1] for r in refs])) and np.array_equal(xys, np.concatenate([r[0] for r in refs]))
    pairs, fit = dd.associateImages([0, 1], [1, 2])
    for s_, d_ in ((0, 1), (1, 2)):
        ep, ef = orc.associate_hamming(refs[s_][1], refs[d_][1], threads=THREADS)
        a, b = int(starts[s_]), int(starts[s_ + 1])
        assert np.array_equal(pairs[a:b], ep) and np.array_equal(fit[a:b], ef)
    assert np.all(pairs[int(starts[2]):] == -1)   # image 2 is no source
    # same frames resident on the device, on torch's stream
    t = torch.from_numpy(np.stack([g.array() for g in imgs])).to("cuda:0")
    ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
    dd2 = fuse(api, ctx=ctx)
    dd2.detectDevice(t.data_ptr(), 300 * 400, 400, 400, 300, 3)
    for i, (pts, words) in enumerate(refs):
        r = dd2._results(i)
        assert np.array_equal(r[0], pts) and np.array_equal(r[3], words)
    ptr, wpf, n = dd2.deviceViewBrief(1)
    assert wpf == 16 and n == len(refs[1][0]) and ptr
    # the words can be handed to the stage-level Hamming entry point as device pointers too: same answer
    with pytest.raises(api.IllegalArgumentException):
        dd2.associateImages([0, 0], [1, 2])   # an image may be the source of one problem per call
    dd2.close(); ctx.close()


@pytest.mark.gpu
def test_fusion_chunked_host_batch_and_config(api, orc, monkeypatch):
    """the chunked host path (uploads under the kernels) gives the plain path's result; a non-default detector configuration goes through"""
    sp, cp = orc.brief_definition()
    imgs = [orc.noise_image(200, 150, 500 + i) for i in range(7)]
    monkeypatch.setenv("BHIP_SURF_CHUNK", "2")
    dd = fuse(api)
    dd.detectBatch([G(api, g) for g in imgs])
    for i, g in enumerate(imgs):
        pts, words = reference(orc, g, sp, cp)
        r = dd._results(i)
        assert np.array_equal(r[0], pts) and np.array_equal(r[3], words)
    monkeypatch.delenv("BHIP_SURF_CHUNK")
    cfg = api.ConfigFastHessian(detectThreshold=5.0, extractRadius=3, maxFeaturesPerScale=-1, initialSampleSize=2, initialSize=9, numberScalesPerOctave=4,
                                numberOfOctaves=3)
    dd = fuse(api, cfg)
    dd.detect(G(api, imgs[0]))
    ocfg = orc.FhCfg(detectThreshold=5.0, extractRadius=3, initialSampleSize=2, numberOfOctaves=3)
    pts = orc.fh_detect(orc.integral(imgs[0]), ocfg, threads=THREADS)
    assert np.array_equal(dd._results(0)[0], pts)
    assert np.array_equal(dd._results(0)[3], orc.brief_describe(imgs[0], pts[:, :2].copy(), 16, sp, cp))


@pytest.mark.gpu
def test_fusion_on_gray_u8(api, orc):
    """GrayU8 frames: GrayS32 integral image for the detector, ImplDescribeBinaryCompare_U8 for the words (its border form shifts the
    word for every pair)"""
    sp, cp = orc.brief_definition()
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, size=(240, 320), dtype=np.uint8)
    dd = fuse(api, imageType=api.GrayU8)
    dd.detect(api.GrayU8(320, 240, frame.reshape(-1).copy()))
    pts = orc.fh_detect_s32(orc.integral_u8(frame), threads=THREADS)
    r = dd._results(0)
    assert len(pts) > 100 and np.array_equal(r[0], pts)
    assert np.array_equal(r[3], orc.brief_describe_u8(frame, pts[:, :2].copy(), 16, sp, cp))


@pytest.mark.gpu
def test_fusion_refuses_surf_only_calls(api, orc):
    dd = fuse(api)
    dd.detect(G(api, orc.noise_image(160, 120, 3)))
    from boofcv_amd import _lib
    import ctypes as C
    L = _lib.load()
    n = dd.getNumberOfFeatures()
    desc = np.zeros((max(n, 1), 64))
    assert L.bhip_surf_fetch(dd._h, 0, None, None, None, desc.ctypes.data_as(_lib._dp)) == _lib.BHIP_ERR_INVALID
    p = np.zeros(max(n, 1), dtype=np.int32); f = np.zeros(max(n, 1))
    a = np.zeros(1, dtype=np.int32)
    assert L.bhip_assoc_l2_surf(dd._h, 1, a.ctypes.data_as(_lib._ip), a.ctypes.data_as(_lib._ip), 1e300, 1, p.ctypes.data_as(_lib._ip), f.ctypes.data_as(_lib._dp)) == _lib.BHIP_ERR_INVALID
    surf = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32)
    surf.detect(G(api, orc.noise_image(160, 120, 3)))
    w = np.zeros((max(surf.getNumberOfFeatures(), 1), 16), dtype=np.int32)
    assert L.bhip_surf_fetch_brief(surf._h, 0, w.ctypes.data_as(_lib._i32p)) == _lib.BHIP_ERR_INVALID


@pytest.mark.gpu
@pytest.mark.parametrize("words", [16, 5])
def test_batched_hamming_association_edge_cases(api, orc, words):
    """bhip_assoc_hamming_dev_batched against the oracle's greedy Hamming association, problem by problem: ragged sizes incl. empty
    source / destination sets, exact duplicates (ties: largest index wins forward, a tie in a column kills the match), cutting and
    degenerate thresholds, with and without backwards validation"""
    import ctypes as C
    import torch
    from boofcv_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(11 + words)
    sizes = [(300, 280), (1, 1), (0, 40), (37, 0), (513, 257), (64, 64)]
    sets = []
    for ns, nd in sizes:
        a = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(ns, words), dtype=np.int64).astype(np.int32)
        b = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(nd, words), dtype=np.int64).astype(np.int32)
        k = min(ns, nd) // 2
        if k:
            b[:k] = a[:k]                      # planted exact matches
            b[k // 2] = b[0]                   # duplicate destination: forward tie -> the larger index
            flip = rng.integers(0, 32, size=k)
            b[np.arange(k), 0] ^= (1 << flip).astype(np.int64).astype(np.int32)   # near matches at distance 1
        if ns > 3:
            a[3] = a[2]                        # duplicate source: a column tie invalidates both
        sets.append((a, b))
    src = np.concatenate([s[0] for s in sets]); dst = np.concatenate([s[1] for s in sets])
    so = np.concatenate([[0], np.cumsum([len(s[0]) for s in sets])]).astype(np.int64)
    do_ = np.concatenate([[0], np.cumsum([len(s[1]) for s in sets])]).astype(np.int64)
    ns = np.array([len(s[0]) for s in sets], dtype=np.int32); nd = np.array([len(s[1]) for s in sets], dtype=np.int32)
    ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
    dsrc = torch.from_numpy(src).to("cuda:0"); ddst = torch.from_numpy(dst if len(dst) else np.zeros((1, words), np.int32)).to("cuda:0")
    LL, I = C.POINTER(C.c_longlong), C.POINTER(C.c_int)
    for maxErr, backwards in ((api.Double_MAX_VALUE, 1), (api.Double_MAX_VALUE, 0), (words * 16 - 20.5, 1), (0.0, 1), (-1.0, 0)):
        pairs = torch.full((len(src),), 7, dtype=torch.int32, device="cuda:0"); fit = torch.zeros(len(src), dtype=torch.float64, device="cuda:0")
        st = L.bhip_assoc_hamming_dev_batched(ctx._h, C.c_void_p(dsrc.data_ptr()), C.c_void_p(ddst.data_ptr()), words, len(sets), so[:-1].copy().ctypes.data_as(LL),
                                              ns.ctypes.data_as(I), do_[:-1].copy().ctypes.data_as(LL), nd.ctypes.data_as(I), maxErr, backwards,
                                              C.c_void_p(pairs.data_ptr()), C.c_void_p(fit.data_ptr()))
        assert st == 0, L.bhip_last_error(ctx._h)
        gp, gf = pairs.cpu().numpy(), fit.cpu().numpy()
        for p, (a, b) in enumerate(sets):
            ep, ef = orc.associate_hamming(a, b, maxErr, bool(backwards), threads=THREADS)
            assert np.array_equal(gp[so[p]:so[p + 1]], ep), (p, maxErr, backwards)
            assert np.array_equal(gf[so[p]:so[p + 1]], ef), (p, maxErr, backwards)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("radius,npts", [(16, 512), (9, 70), (3, 33), (20, 600)])
def test_brief_patch_and_gather_kernels_agree_with_the_oracle(api, orc, monkeypatch, radius, npts):
    """the LDS-patch kernel (one wave per point) and the per-word gather kernel (BHIP_BRIEF_GATHER=1, also the fall-back for definitions whose
    samples leave the patch) against the oracle: word counts that are no multiple of 32 (the last word is right aligned), interior and
    border points, GrayF32 (skip without shift at the border) and GrayU8 (shift for every pair)"""
    rng = np.random.default_rng(radius * 1000 + npts)
    sp = rng.integers(-radius, radius + 1, size=(npts, 2)).astype(np.int32)
    cp = np.stack([np.arange(npts), rng.integers(0, npts, size=npts)], axis=1).astype(np.int32)
    w, h = 190, 141
    img = orc.noise_image(w, h, 5)
    xy = np.concatenate([rng.uniform(0, w, size=(300, 1)), rng.uniform(0, h, size=(300, 1))], axis=1)
    xy = np.concatenate([xy, [[0, 0], [w - 0.1, h - 0.1], [radius, radius], [radius - 0.1, 70], [w - 1 - radius, h - 1 - radius], [w - radius, h - radius]]])
    ref = orc.brief_describe(img, xy, radius, sp, cp)
    u8 = (img.array() * 2.5).astype(np.uint8)
    ref8 = orc.brief_describe_u8(u8, xy, radius, sp, cp)
    for gather in (False, True):
        if gather:
            monkeypatch.setenv("BHIP_BRIEF_GATHER", "1")
        b = api.DescribePointBrief(radius, sp, cp); b.setImage(G(api, img))
        assert np.array_equal(b.processAll(xy), ref), "F32 gather=%s" % gather
        b.setImage(api.GrayU8(w, h, u8.reshape(-1).copy()))
        assert np.array_equal(b.processAll(xy), ref8), "U8 gather=%s" % gather
    monkeypatch.delenv("BHIP_BRIEF_GATHER")
    # samples outside [-radius, radius]: the library must still answer (gather kernel), as the reference does for any definition
    sp2 = sp.copy(); sp2[0] = (radius + 3, 0)
    b = api.DescribePointBrief(radius, sp2, cp); b.setImage(G(api, img))
    assert np.array_equal(b.processAll(xy), orc.brief_describe(img, xy, radius, sp2, cp))
