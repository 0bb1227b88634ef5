"""Host-side mirror of the reference's interfaces for the detect -> describe -> associate path, on top of the C ABI.

Names, argument meaning and error behaviour follow the Java reference (waicool20/BoofCV 0.35-SNAPSHOT) so that the parity
tests read like the reference's own tests; every call runs hand-written HIP kernels in libboofhip.so on an MI355X.

    F: = main/boofcv-feature/src/main/java/boofcv/   I: = main/boofcv-ip/src/main/java/boofcv/   T: = main/boofcv-types/src/main/java/boofcv/

  FactoryDetectDescribe.surfStable / surfFast   F:factory/feature/detdesc/FactoryDetectDescribe.java:118-135,209-226
  DetectDescribePoint                            F:abst/feature/detdesc/DetectDescribePoint.java:32-46
  FactoryAssociation.greedy / AssociateDescription   F:factory/feature/associate/FactoryAssociation.java:51-65 ; F:abst/feature/associate/AssociateDescription.java:42-61
  BOverride* hooks (static op classes)           I:alg/filter/convolve/BOverrideConvolveImage.java:37-83 etc.

Errors: BHIP_ERR_INVALID -> IllegalArgumentException, everything else -> RuntimeError (which is what a BOverride hook throws to make the
reference fall back to its Java code).
"""
import atexit
import ctypes as C
import math
import sys
import weakref
from dataclasses import dataclass

import numpy as np

from . import _lib

Double_MAX_VALUE = 1.7976931348623157e308
Float_MAX_VALUE = float(np.finfo(np.float32).max)


class IllegalArgumentException(ValueError):
    pass


def _check(ctx, status):
    if status == _lib.BHIP_OK:
        return
    msg = _lib.load().bhip_last_error(ctx._h if ctx is not None else None)
    msg = msg.decode(errors="replace") if msg else ""
    if status == _lib.BHIP_ERR_INVALID:
        raise IllegalArgumentException(msg or "invalid argument")
    raise RuntimeError("boofhip status %d: %s" % (status, msg))


class Context:
    """bhip_ctx: one per host thread per device.

    Lifetime: close() (or garbage collection) destroys the native context after closing every detect+describe object created on it; the
    native library tolerates any order anyway (include/boofhip.h, "handles may be destroyed in any order").  At interpreter exit an
    atexit hook closes every live context while the HIP runtime is still up; finalisers that run later do nothing."""
    _default = {}
    _live = weakref.WeakSet()

    def __init__(self, device=0, stream=None):
        L = _lib.load()
        h = C.c_void_p()
        if stream is None:
            st = L.bhip_ctx_create(device, C.byref(h))
        else:
            st = L.bhip_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if st != _lib.BHIP_OK:
            raise RuntimeError("bhip_ctx_create(device=%d) failed with status %d: no usable MI355X? (there is no CPU fallback)" % (device, st))
        self._h = h
        self.device = device
        self._children = weakref.WeakSet()
        Context._live.add(self)

    def synchronize(self):
        _check(self, _lib.load().bhip_ctx_synchronize(self._h))

    def lastError(self):
        msg = _lib.load().bhip_last_error(self._h)
        return msg.decode(errors="replace") if msg else ""

    def profile(self, on=True):
        """Bracket every kernel launch with HIP events on this context's stream."""
        _check(self, _lib.load().bhip_profile_enable(self._h, 1 if on else 0))

    def profileReset(self):
        _check(self, _lib.load().bhip_profile_reset(self._h))

    def profileReport(self):
        """{tag: dict(launches, ms, bytes, flops)} accumulated since the last reset."""
        L = _lib.load()
        n = L.bhip_profile_report(self._h, None, 0)
        buf = C.create_string_buffer(max(n, 1))
        L.bhip_profile_report(self._h, buf, n)
        out = {}
        for line in buf.value.decode().splitlines():
            tag, launches, ms, b, f = line.split()
            out[tag] = dict(launches=int(launches), ms=float(ms), bytes=float(b), flops=float(f))
        return out

    def close(self):
        if self._h:
            for child in list(self._children):
                child.close()
            _lib.load().bhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self, _finalizing=sys.is_finalizing):   # (bound at definition: module globals are gone when this runs late in shutdown)
        try:
            if _finalizing():
                return   # the atexit hook below has closed what was alive; the native library ignores destroy calls after exit began
            self.close()
        except Exception:
            pass

    @classmethod
    def _close_all(cls):
        for ctx in list(cls._live):
            try:
                ctx.close()
            except Exception:
                pass
        cls._default.clear()

    @classmethod
    def default(cls, device=0):
        if device not in cls._default:
            cls._default[device] = Context(device)
        return cls._default[device]


atexit.register(Context._close_all)   # runs before module teardown and before the HIP runtime's own exit handlers


class _PinnedPool:
    """Page-locked host blocks (bhip_host_alloc) for the arrays this wrapper hands out or fills on every call -- fetched key points and
    descriptors, match lists -- so that their copies are DMA transfers instead of staged pageable copies (what a JNI provider gets from direct
    ByteBuffers over the same allocator).  Blocks are recycled by size class when the numpy arrays built on them are garbage collected."""
    _free = {}      # size class -> [address]
    _pooled = 0     # bytes sitting in _free
    _closed = False
    MIN = 4096
    MAX_POOLED = 1 << 30   # blocks released beyond this go back to the runtime instead of the pool

    @classmethod
    def _size_class(cls, nbytes):
        c = cls.MIN   # 4K, 6K, 8K, 12K, 16K, 24K, ...: powers of two and the sizes half way between them
        while c < nbytes:
            c = c * 3 // 2 if c & (c - 1) == 0 else c * 4 // 3
        return c

    @classmethod
    def block(cls, ctx, nbytes):
        """-> (ctypes uint8 array over a pinned block of at least nbytes, or None when pinned memory is not to be had)"""
        if cls._closed or not ctx._h:
            return None
        size = cls._size_class(max(int(nbytes), 1))
        addr = None
        lst = cls._free.get(size)
        if lst:
            try:
                addr = lst.pop()      # (list.pop is atomic; another thread may have taken the last block in between)
                cls._pooled -= size
            except IndexError:
                addr = None
        if addr is None:
            p = C.c_void_p()
            if _lib.load().bhip_host_alloc(ctx._h, size, C.byref(p)) != _lib.BHIP_OK or not p.value:
                return None
            addr = p.value
        buf = (C.c_uint8 * size).from_address(addr)
        weakref.finalize(buf, cls._release, addr, size)
        return buf

    @classmethod
    def _release(cls, addr, size, _finalizing=sys.is_finalizing):
        if cls._closed or _finalizing():
            return   # the exit hook has run (or the interpreter is going down): the runtime reclaims the block
        if cls._pooled + size > cls.MAX_POOLED:
            _lib.load().bhip_host_free(C.c_void_p(addr))
            return
        cls._free.setdefault(size, []).append(addr)
        cls._pooled += size

    @classmethod
    def arrays(cls, ctx, specs):
        """specs = [(shape, dtype), ...] -> numpy arrays carved out of ONE pinned block (64-byte aligned each); pageable arrays when no pinned
        memory is available.  Contents are uninitialised."""
        sizes = [int(np.prod(shape)) * np.dtype(dt).itemsize for shape, dt in specs]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) & ~63
        buf = cls.block(ctx, total) if total else None
        if buf is None:
            return [np.empty(shape, dtype=dt) for shape, dt in specs]
        raw = np.frombuffer(buf, dtype=np.uint8)   # keeps `buf` (and with it the block) alive through .base
        return [raw[o:o + n].view(dt).reshape(shape) for o, n, (shape, dt) in zip(offs, sizes, specs)]

    @classmethod
    def _close(cls):
        cls._closed = True
        L = _lib.load()
        for lst in cls._free.values():
            for addr in lst:
                L.bhip_host_free(C.c_void_p(addr))
        cls._free.clear()
        cls._pooled = 0


atexit.register(_PinnedPool._close)   # registered after Context._close_all, so it runs before it (contexts are still alive)


# ------------------------------------------------------------------------------------------------------------------
# data types
# ------------------------------------------------------------------------------------------------------------------
def _check_extent(width, height, startIndex, stride, size):
    """The raw pointer goes to the C ABI: the view must lie inside its array (a Java array access would throw instead of reading past the end)."""
    if width < 0 or height < 0 or startIndex < 0 or stride < width:
        raise IllegalArgumentException("bad image geometry: width %d height %d startIndex %d stride %d" % (width, height, startIndex, stride))
    if width > 0 and height > 0 and startIndex + (height - 1) * stride + width > size:
        raise IllegalArgumentException("image view (startIndex %d, stride %d, %d x %d) exceeds its data array of %d elements" % (startIndex, stride, width, height, size))


class GrayF32:
    """T:struct/image/GrayF32.java:30 / ImageBase.java:34-52: pixel (x,y) = data[startIndex + y*stride + x]."""

    def __init__(self, width=0, height=0, data=None, startIndex=0, stride=None):
        self.width, self.height = int(width), int(height)
        self.stride = int(width if stride is None else stride)
        self.startIndex = int(startIndex)
        if data is None:
            data = np.zeros(self.startIndex + self.stride * self.height, dtype=np.float32)
        if data.dtype != np.float32 or not data.flags["C_CONTIGUOUS"] or data.ndim != 1:
            raise IllegalArgumentException("data must be a contiguous 1-D float32 array")
        _check_extent(self.width, self.height, self.startIndex, self.stride, data.size)
        self.data = data

    @staticmethod
    def wrap(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return GrayF32(a.shape[1], a.shape[0], a.reshape(-1))

    def reshape(self, width, height):
        if width * height > self.data.size or self.startIndex != 0:
            self.data = np.zeros(width * height, dtype=np.float32)
            self.startIndex = 0
        self.width, self.height, self.stride = int(width), int(height), int(width)

    def subimage(self, x0, y0, x1, y1):
        if not (0 <= x0 <= x1 <= self.width and 0 <= y0 <= y1 <= self.height):   # ImageBase.subimage throws IllegalArgumentException
            raise IllegalArgumentException("sub-image (%d,%d)-(%d,%d) is outside the %d x %d image" % (x0, y0, x1, y1, self.width, self.height))
        return GrayF32(x1 - x0, y1 - y0, self.data, self.startIndex + y0 * self.stride + x0, self.stride)

    def array(self):
        return np.lib.stride_tricks.as_strided(self.data[self.startIndex:], shape=(self.height, self.width), strides=(4 * self.stride, 4))

    def get(self, x, y):
        if not (0 <= x < self.width and 0 <= y < self.height):
            raise IndexError("Requested pixel is out of bounds: %d %d" % (x, y))  # ImageAccessException
        return float(self.data[self.startIndex + y * self.stride + x])

    def set(self, x, y, v):
        if not (0 <= x < self.width and 0 <= y < self.height):
            raise IndexError("Requested pixel is out of bounds: %d %d" % (x, y))
        self.data[self.startIndex + y * self.stride + x] = v

    def _p(self):
        return self.data.ctypes.data_as(C.POINTER(C.c_float))


@dataclass
class Point2D_F64:
    x: float = 0.0
    y: float = 0.0


@dataclass
class Point2D_I16:
    x: int = 0
    y: int = 0


class TupleDesc_F64:
    """F:struct/feature/TupleDesc_F64.java:30"""

    def __init__(self, numFeatures=0, value=None):
        self.value = np.zeros(numFeatures, dtype=np.float64) if value is None else np.asarray(value, dtype=np.float64)

    def size(self):
        return len(self.value)

    def setTo(self, src):
        self.value = np.array(src.value, dtype=np.float64)


class _GrayInt:
    """Integer single-band images (T:struct/image/GrayU8.java, GrayS32.java): pixel (x,y) = data[startIndex + y*stride + x]."""
    dtype = None

    def __init__(self, width=0, height=0, data=None, startIndex=0, stride=None):
        self.width, self.height = int(width), int(height)
        self.stride = int(width if stride is None else stride)
        self.startIndex = int(startIndex)
        if data is None:
            data = np.zeros(self.startIndex + self.stride * self.height, dtype=self.dtype)
        if data.dtype != self.dtype or not data.flags["C_CONTIGUOUS"] or data.ndim != 1:
            raise IllegalArgumentException("data must be a contiguous 1-D %s array" % np.dtype(self.dtype).name)
        _check_extent(self.width, self.height, self.startIndex, self.stride, data.size)
        self.data = data

    @classmethod
    def wrap(cls, a):
        a = np.ascontiguousarray(a, dtype=cls.dtype)
        return cls(a.shape[1], a.shape[0], a.reshape(-1))

    def reshape(self, width, height):
        if width * height > self.data.size or self.startIndex != 0:
            self.data = np.zeros(width * height, dtype=self.dtype)
            self.startIndex = 0
        self.width, self.height, self.stride = int(width), int(height), int(width)

    def subimage(self, x0, y0, x1, y1):
        if not (0 <= x0 <= x1 <= self.width and 0 <= y0 <= y1 <= self.height):
            raise IllegalArgumentException("sub-image (%d,%d)-(%d,%d) is outside the %d x %d image" % (x0, y0, x1, y1, self.width, self.height))
        return type(self)(x1 - x0, y1 - y0, self.data, self.startIndex + y0 * self.stride + x0, self.stride)

    def array(self):
        it = np.dtype(self.dtype).itemsize
        return np.lib.stride_tricks.as_strided(self.data[self.startIndex:], shape=(self.height, self.width), strides=(it * self.stride, it))


class GrayU8(_GrayInt):
    dtype = np.uint8

    def _p(self):
        return self.data.ctypes.data_as(_lib._u8p)


class GrayS32(_GrayInt):
    dtype = np.int32

    def _p(self):
        return self.data.ctypes.data_as(_lib._i32p)


class Planar:
    """T:struct/image/Planar.java: bands of one shape.  Planar(GrayF32, width, height, numBands) or Planar.wrap([bands])."""

    def __init__(self, bandType=None, width=0, height=0, numBands=0):
        if bandType is not None and bandType is not GrayF32:
            raise RuntimeError("only GrayF32 bands are implemented on the GPU")
        self.width, self.height = int(width), int(height)
        self.bands = [GrayF32(width, height) for _ in range(numBands)]

    @staticmethod
    def wrap(bands):
        p = Planar(GrayF32, bands[0].width, bands[0].height, 0)
        for b in bands:
            if b.width != p.width or b.height != p.height:
                raise IllegalArgumentException("bands must have the same shape")
        p.bands = list(bands)
        return p

    def getNumBands(self):
        return len(self.bands)

    def getBand(self, i):
        return self.bands[i]


class PlanarType:
    """ImageType.pl(numBands, GrayF32.class)"""

    def __init__(self, numBands, bandType=None):
        self.numBands, self.bandType = int(numBands), bandType or GrayF32


class BrightFeature(TupleDesc_F64):
    """F:struct/feature/BrightFeature.java:32: SURF descriptor + sign of the Laplacian."""

    def __init__(self, numFeatures=0, value=None, white=False):
        super().__init__(numFeatures, value)
        self.white = bool(white)

    def setTo(self, src):
        super().setTo(src)
        self.white = getattr(src, "white", False)


class TupleDesc_B:
    """F:struct/feature/TupleDesc_B.java:27-40: numBits packed into int32 words."""

    def __init__(self, numBits, data=None):
        self.numBits = int(numBits)
        n = (self.numBits + 31) // 32
        self.data = np.zeros(n, dtype=np.int32) if data is None else np.asarray(data, dtype=np.int32)


@dataclass
class AssociatedIndex:
    """F:struct/feature/AssociatedIndex.java:30-34"""
    src: int = 0
    dst: int = 0
    fitScore: float = 0.0


class MatchScoreType:
    NORM_ERROR = "NORM_ERROR"


# ------------------------------------------------------------------------------------------------------------------
# configuration (public mutable fields, null => defaults, as in the reference)
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class ConfigFastHessian:
    """F:abst/feature/detect/interest/ConfigFastHessian.java:33-70"""
    detectThreshold: float = 1.0
    extractRadius: int = 2
    maxFeaturesPerScale: int = -1
    initialSampleSize: int = 1
    initialSize: int = 9
    numberScalesPerOctave: int = 4
    numberOfOctaves: int = 4
    scaleStepSize: int = 6

    def _c(self):
        return _lib.FhCfg(self.detectThreshold, self.extractRadius, self.maxFeaturesPerScale, self.initialSampleSize, self.initialSize,
                          self.numberScalesPerOctave, self.numberOfOctaves, self.scaleStepSize)


class ConfigSurfDescribe:
    """F:abst/feature/describe/ConfigSurfDescribe.java:34-78"""

    @dataclass
    class Speed:
        widthLargeGrid: int = 4
        widthSubRegion: int = 5
        widthSample: int = 3
        useHaar: bool = False
        weightSigma: float = 4.5

    @dataclass
    class Stability:
        widthLargeGrid: int = 4
        widthSubRegion: int = 5
        widthSample: int = 3
        useHaar: bool = False
        overLap: int = 2
        sigmaLargeGrid: float = 2.5
        sigmaSubRegion: float = 2.5


@dataclass
class ConfigSlidingIntegral:
    """F:abst/feature/orientation/ConfigSlidingIntegral.java:34-54"""
    objectRadiusToScale: float = 0.5
    samplePeriod: float = 0.65
    windowSize: float = math.pi / 3.0
    radius: int = 8
    weightSigma: float = -1.0
    sampleWidth: int = 6


@dataclass
class ConfigAverageIntegral:
    """F:abst/feature/orientation/ConfigAverageIntegral.java:34-51"""
    objectRadiusToScale: float = 0.5
    radius: int = 6
    samplePeriod: float = 1.0
    sampleWidth: int = 6
    weightSigma: float = -1.0


@dataclass
class ConfigExtract:
    """F:abst/feature/detect/extract/ConfigExtract.java:32-56"""
    radius: int = 1
    threshold: float = 0.0
    ignoreBorder: int = 0
    useStrictRule: bool = True
    detectMinimums: bool = False
    detectMaximums: bool = True

    def checkValidity(self):
        if self.radius <= 0:
            raise IllegalArgumentException("Search radius must be >= 1")
        if self.ignoreBorder < 0:
            raise IllegalArgumentException("Ignore border must be >= 0 ")


# ------------------------------------------------------------------------------------------------------------------
# detect + describe
# ------------------------------------------------------------------------------------------------------------------
class DetectDescribePoint:
    """DetectDescribePoint<GrayF32,BrightFeature> backed by bhip_surf (WrapDetectDescribeSurf.java:47-159).

    Results are recycled on the next detect(), instances are not thread safe -- both as in the reference.
    detectBatch() is the batched extension (one launch sequence for many frames)."""

    def __init__(self, stable, configDetector, configDescribe, configOrientation, ctx=None):
        self.ctx = ctx or Context.default()
        L = _lib.load()
        fh = (configDetector or ConfigFastHessian())._c()
        if stable:
            d = configDescribe or ConfigSurfDescribe.Stability()
            sd = _lib.SurfCfg(d.widthLargeGrid, d.widthSubRegion, d.widthSample, 4.5, d.overLap, d.sigmaLargeGrid, d.sigmaSubRegion)
            o = configOrientation or ConfigSlidingIntegral()
            oc = _lib.OriCfg(o.objectRadiusToScale, o.samplePeriod, o.windowSize, o.radius, o.weightSigma, o.sampleWidth)
        else:
            d = configDescribe or ConfigSurfDescribe.Speed()
            sd = _lib.SurfCfg(d.widthLargeGrid, d.widthSubRegion, d.widthSample, d.weightSigma, 2, 2.5, 2.5)
            o = configOrientation or ConfigAverageIntegral()
            oc = _lib.OriCfg(o.objectRadiusToScale, o.samplePeriod, 0.0, o.radius, o.weightSigma, o.sampleWidth)
        if d.useHaar:
            raise RuntimeError("useHaar=true is not implemented on the GPU (use the Java path)")
        h = C.c_void_p()
        _check(self.ctx, L.bhip_surf_create(self.ctx._h, C.byref(fh), C.byref(sd), C.byref(oc), 1 if stable else 0, C.byref(h)))
        self._h = h
        self.ctx._children.add(self)
        self._dof = L.bhip_surf_dof(h)
        self._batch = 0
        self._image = 0
        self._cache = {}

    def close(self):
        """Releases the native object (idempotent; safe after its context has been closed)."""
        if self._h:
            _lib.load().bhip_surf_destroy(self._h)
            self._h = None

    def __del__(self, _finalizing=sys.is_finalizing):
        try:
            if _finalizing():
                return
            self.close()
        except Exception:
            pass

    # --- DescriptorInfo
    def createDescription(self):
        return BrightFeature(self._dof)

    def getDescriptionType(self):
        return BrightFeature

    # --- detection
    def detect(self, input):
        self.detectBatch([input])

    def detectBatch(self, images):
        if not images:
            raise IllegalArgumentException("empty batch")
        w, h = images[0].width, images[0].height
        for im in images:
            if im.width != w or im.height != h:
                raise IllegalArgumentException("all images of a batch must have the same shape")
        for im in images:
            _check_extent(im.width, im.height, im.startIndex, im.stride, im.data.size)   # the fields are mutable: validate what is handed over
        n = len(images)
        u8 = isinstance(images[0], GrayU8)
        if any(isinstance(im, GrayU8) != u8 for im in images):
            raise IllegalArgumentException("all images of a batch must have the same type")
        ptrs = (C.POINTER(C.c_uint8 if u8 else C.c_float) * n)(*[im._p() for im in images])
        starts = (C.c_int * n)(*[im.startIndex for im in images])
        strides = (C.c_int * n)(*[im.stride for im in images])
        self._cache = {}
        self._batch = 0
        fn = _lib.load().bhip_surf_detect_u8 if u8 else _lib.load().bhip_surf_detect_f32
        _check(self.ctx, fn(self._h, ptrs, starts, strides, w, h, n))
        self._batch = n
        self._image = 0
        self._shape = (w, h)

    def detectDevice(self, dev_ptr, imageStride, stride, width, height, batch):
        """Batch already resident in HBM (bench path): dev_ptr is a device address of float32 pixels."""
        self._cache = {}
        self._batch = 0
        _check(self.ctx, _lib.load().bhip_surf_detect_dev_f32(self._h, C.c_void_p(dev_ptr), imageStride, stride, width, height, batch))
        self._batch = batch
        self._image = 0
        self._shape = (width, height)

    def selectImage(self, image):
        """Which image of the last batch the index-based getters refer to (0 for the single-image reference call)."""
        if not (0 <= image < self._batch):
            raise IllegalArgumentException("image index out of range")
        self._image = image

    def _results(self, image=None):
        image = self._image if image is None else image
        if image not in self._cache:
            L = _lib.load()
            n = C.c_int(0)
            _check(self.ctx, L.bhip_surf_count(self._h, image, C.byref(n)))
            n = n.value
            # page-locked result arrays (the copies are DMA transfers; a descriptor list handed on to associate() uploads the same way)
            xys, ang, white, desc = _PinnedPool.arrays(self.ctx, [((n, 3), np.float64), ((n,), np.float64), ((n,), np.uint8), ((n, self._dof), np.float64)])
            if n:
                _check(self.ctx, L.bhip_surf_fetch(self._h, image, xys.ctypes.data_as(_lib._dp), ang.ctypes.data_as(_lib._dp),
                                                   white.ctypes.data_as(_lib._u8p), desc.ctypes.data_as(_lib._dp)))
            self._cache[image] = (xys, ang, white, desc)
        return self._cache[image]

    def fetchAll(self, out=None):
        """The whole batch of the last detect in one set of copies (bhip_surf_fetch_all): (xy_scale [total,3], angle [total],
        white [total], desc [total,dof], starts [batch+1]); image i owns rows starts[i]:starts[i+1].
        out = (xy_scale, angle, white, desc) receives the copies when given: C-contiguous float64 / uint8 arrays with at least `total` rows
        (e.g. views of page-locked memory a caller keeps across batches -- the copies then run at PCIe speed); views of the first `total`
        rows are returned."""
        L = _lib.load()
        counts = self.counts()
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        total = int(starts[-1])
        if out is None:
            xys = np.empty((total, 3)); ang = np.empty(total); white = np.empty(total, dtype=np.uint8); desc = np.empty((total, self._dof))
        else:
            xys, ang, white, desc = out
            want = ((xys, np.float64, (3,)), (ang, np.float64, ()), (white, np.uint8, ()), (desc, np.float64, (self._dof,)))
            for a, dt, tail in want:
                if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous and a.shape[1:] == tail and a.shape[0] >= total):
                    raise IllegalArgumentException("fetchAll: output arrays must be C-contiguous, of the right type and at least %d rows long" % total)
            xys, ang, white, desc = xys[:total], ang[:total], white[:total], desc[:total]
        if total:
            _check(self.ctx, L.bhip_surf_fetch_all(self._h, xys.ctypes.data_as(_lib._dp), ang.ctypes.data_as(_lib._dp), white.ctypes.data_as(_lib._u8p),
                                                   desc.ctypes.data_as(_lib._dp)))
        return xys, ang, white, desc, starts

    def associateImages(self, srcImages, dstImages, maxError=Double_MAX_VALUE, backwardsValidation=True):
        """Greedy Euclidean-squared association of image srcImages[p] with image dstImages[p] of the last detect, on the descriptors still
        resident on the device (bhip_assoc_l2_surf).  -> (pairs, fitQuality) over the batch's compact key-point index space (see fetchAll)."""
        L = _lib.load()
        a = np.ascontiguousarray(srcImages, dtype=np.int32)
        b = np.ascontiguousarray(dstImages, dtype=np.int32)
        if a.shape != b.shape:
            raise IllegalArgumentException("source and destination image lists differ in length")
        total = max(self.totalFeatures(), 1)
        pairs = np.full(total, -1, dtype=np.int32)
        fit = np.zeros(total)
        _check(self.ctx, L.bhip_assoc_l2_surf(self._h, len(a), a.ctypes.data_as(_lib._ip), b.ctypes.data_as(_lib._ip), float(maxError),
                                              1 if backwardsValidation else 0, pairs.ctypes.data_as(_lib._ip), fit.ctypes.data_as(_lib._dp)))
        return pairs, fit

    def counts(self):
        """getNumberOfFeatures() of every image of the last batch (one native call) -> int32 array"""
        out = np.zeros(max(self._batch, 1), dtype=np.int32)
        if self._batch:
            _check(self.ctx, _lib.load().bhip_surf_counts(self._h, out.ctypes.data_as(_lib._ip), len(out)))
        return out[:self._batch]

    def totalFeatures(self):
        n = C.c_longlong(0)
        _check(self.ctx, _lib.load().bhip_surf_total(self._h, C.byref(n)))
        return n.value

    def deviceView(self, image):
        """(dev_desc_ptr, dev_keypoint_ptr, dev_white_ptr, n) of image `image` -- valid until the next detect."""
        d, k, w, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        _check(self.ctx, _lib.load().bhip_surf_dev_view(self._h, image, C.byref(d), C.byref(k), C.byref(w), C.byref(n)))
        return d.value, k.value, w.value, n.value

    def describePoints(self, xy_scale, image=0):
        """computeDescriptors() for a caller-supplied point list on the integral image of the last detect."""
        pts = np.ascontiguousarray(xy_scale, dtype=np.float64).reshape(-1, 3)
        n = len(pts)
        ang = np.zeros(n); white = np.zeros(n, dtype=np.uint8); desc = np.zeros((n, self._dof))
        _check(self.ctx, _lib.load().bhip_surf_describe_points(self._h, image, pts.ctypes.data_as(_lib._dp), n, ang.ctypes.data_as(_lib._dp),
                                                              white.ctypes.data_as(_lib._u8p), desc.ctypes.data_as(_lib._dp)))
        return ang, white, desc

    def fetchIntegral(self, image, width=None, height=None):
        """Integral image of image `image` of the last detect.  The buffer is sized from the detector's own shape (the C side copies
        W*H words of the last detect); width / height, when given, must agree with it."""
        w, h = self._shape
        if (width is not None and width != w) or (height is not None and height != h):
            raise IllegalArgumentException("the last detect ran on %d x %d images, not %s x %s" % (w, h, width, height))
        out = np.zeros((h, w), dtype=np.float32)
        _check(self.ctx, _lib.load().bhip_surf_fetch_integral(self._h, image, out.ctypes.data_as(_lib._fp)))
        return out

    # --- InterestPointDetector / FoundPointSO
    def getNumberOfFeatures(self):
        return len(self._results()[0])

    def getLocation(self, featureIndex):
        x = self._results()[0][featureIndex]
        return Point2D_F64(float(x[0]), float(x[1]))

    def getRadius(self, featureIndex):
        return float(self._results()[0][featureIndex][2]) * 2.0  # BoofDefaults.SURF_SCALE_TO_RADIUS

    def getOrientation(self, featureIndex):
        return float(self._results()[1][featureIndex])

    def getDescription(self, index):
        r = self._results()
        return BrightFeature(self._dof, r[3][index], bool(r[2][index]))

    def hasScale(self):
        return True

    def hasOrientation(self):
        return True


class SurfPlanar_to_DetectDescribePoint(DetectDescribePoint):
    """DetectDescribePoint<Planar<GrayF32>,BrightFeature> (F:abst/feature/detdesc/SurfPlanar_to_DetectDescribePoint.java:40-132 over
    F:alg/feature/detdesc/DetectDescribeSurfPlanar.java and F:alg/feature/describe/DescribePointSurfPlanar.java): key points from the band
    average, orientation with object radius = scale, one descriptor per band concatenated and normalised as a whole."""

    def __init__(self, stable, configDetector, configDescribe, configOrientation, numBands, ctx=None):
        super().__init__(stable, configDetector, configDescribe, configOrientation, ctx)
        self.numBands = int(numBands)
        self._dof = self._dof * self.numBands

    def detect(self, input):
        if input.getNumBands() != self.numBands:
            raise IllegalArgumentException("Unexpected number of bands. Expected %d found %d" % (self.numBands, input.getNumBands()))
        b0 = input.getBand(0)
        for b in input.bands:
            if (b.startIndex, b.stride) != (b0.startIndex, b0.stride):
                raise IllegalArgumentException("bands must share startIndex and stride")   # Planar images do (ImageMultiBand layout)
        ptrs = (C.POINTER(C.c_float) * self.numBands)(*[b._p() for b in input.bands])
        self._cache = {}
        self._batch = 0
        for b in input.bands:
            _check_extent(input.width, input.height, b.startIndex, b.stride, b.data.size)
        _check(self.ctx, _lib.load().bhip_surf_detect_planar_f32(self._h, ptrs, self.numBands, b0.startIndex, b0.stride, input.width, input.height))
        self._batch = 1
        self._image = 0
        self._shape = (input.width, input.height)

    def detectBatch(self, images):
        raise RuntimeError("colour SURF processes one planar frame per call")

    def getRadius(self, featureIndex):
        return float(self._results()[0][featureIndex][2])   # DetectDescribeSurfPlanar.getRadius: the scale itself


class Random:
    """java.util.Random (the JDK's documented linear congruential generator; host-side, needed only to build the BRIEF definition the way
    FactoryBriefDefinition does).  nextGaussian uses math.log / math.sqrt where Java uses StrictMath: the table is *parity unpinned* in the
    last ulp of log -- a Java caller passes the definition its own JVM generated."""

    def __init__(self, seed):
        self.seed = (int(seed) ^ 0x5DEECE66D) & ((1 << 48) - 1)
        self._next_gaussian = None

    def next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.seed >> (48 - bits)
        return v - (1 << 32) if v >= (1 << 31) and bits == 32 else v

    def nextInt(self, bound=None):
        if bound is None:
            return self.next(32)
        if bound <= 0:
            raise IllegalArgumentException("bound must be positive")
        if (bound & -bound) == bound:
            return (bound * self.next(31)) >> 31
        while True:
            bits = self.next(31)
            val = bits % bound
            if bits - val + (bound - 1) < (1 << 31):
                return val

    def nextDouble(self):
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))

    def nextGaussian(self):
        if self._next_gaussian is not None:
            g, self._next_gaussian = self._next_gaussian, None
            return g
        while True:
            v1 = 2 * self.nextDouble() - 1
            v2 = 2 * self.nextDouble() - 1
            s = v1 * v1 + v2 * v2
            if 0 < s < 1:
                break
        m = math.sqrt(-2 * math.log(s) / s)
        self._next_gaussian = v2 * m
        return v1 * m


class BinaryCompareDefinition_I32:
    """F:alg/feature/describe/brief/BinaryCompareDefinition_I32.java: sample points (x,y) and the index pairs that are compared."""

    def __init__(self, radius, samplePoints, compare):
        self.radius = int(radius)
        self.samplePoints = np.ascontiguousarray(samplePoints, dtype=np.int32).reshape(-1, 2)
        self.compare = np.ascontiguousarray(compare, dtype=np.int32).reshape(-1, 2)

    def getLength(self):
        return len(self.compare)


class FactoryBriefDefinition:
    @staticmethod
    def gaussian2(rand, radius, numPairs):
        """F:alg/feature/describe/brief/FactoryBriefDefinition.java:57-85: sample i = (int)(gaussian * sigma) per axis, redrawn until it lies
        inside the circle; compare[i] = (i, rand.nextInt(numPairs)); the RNG calls interleave exactly as in the reference."""
        sigma = (2.0 * radius + 1.0) / 5.0
        pts = np.zeros((numPairs, 2), dtype=np.int32)
        cmp_ = np.zeros((numPairs, 2), dtype=np.int32)
        for i in range(numPairs):
            while True:
                x = int(rand.nextGaussian() * sigma)
                y = int(rand.nextGaussian() * sigma)
                if math.sqrt(x * x + y * y) < radius:
                    break
            pts[i] = (x, y)
            cmp_[i] = (i, rand.nextInt(numPairs))
        return BinaryCompareDefinition_I32(radius, pts, cmp_)


@dataclass
class ConfigBrief:
    """F:abst/feature/describe/ConfigBrief.java:33-52"""
    radius: int = 16
    numPoints: int = 512
    blurSigma: float = -1
    blurRadius: int = 4
    fixed: bool = True

    def checkValidity(self):
        pass


class WrapDescribeBrief:
    """DescribeRegionPoint<T,TupleDesc_B> (F:abst/feature/describe/WrapDescribeBrief.java:30-86) over DescribePointBrief: process() ignores
    orientation and radius and always succeeds."""

    def __init__(self, definition, imageType, ctx=None):
        self.definition = definition
        self.imageType = imageType
        self.length = definition.getLength()
        self.alg = DescribePointBrief(definition.radius, definition.samplePoints, definition.compare, ctx=ctx)

    def createDescription(self):
        return TupleDesc_B(self.length)

    def setImage(self, image):
        self.alg.setImage(image)

    def process(self, x, y, orientation, radius, storage):
        self.alg.process(x, y, storage)
        return True

    def requiresRadius(self): return False
    def requiresOrientation(self): return False
    def getImageType(self): return self.imageType
    def getDescriptionType(self): return TupleDesc_B
    def getCanonicalWidth(self): return self.definition.radius * 2 + 1


class FactoryDescribeRegionPoint:
    @staticmethod
    def brief(config=None, imageType=GrayF32, definition=None, ctx=None):
        """F:factory/feature/describe/FactoryDescribeRegionPoint.java:187-202.  `definition` lets a caller hand in the table its JVM made
        (FactoryBriefDefinition.gaussian2(new Random(123), radius, numPoints)); otherwise it is generated here the same way."""
        config = config or ConfigBrief()
        config.checkValidity()
        if not config.fixed:
            raise RuntimeError("the scale / orientation aware BRIEF (WrapDescribeBriefSo) is not implemented on the GPU (use the Java path)")
        if imageType is not GrayF32 and imageType is not GrayU8:
            raise RuntimeError("only GrayF32 and GrayU8 are implemented on the GPU (use the Java path)")
        if definition is None:
            definition = FactoryBriefDefinition.gaussian2(Random(123), config.radius, config.numPoints)
        return WrapDescribeBrief(definition, imageType, ctx=ctx)


class WrapFHtoInterestPoint:
    """InterestPointDetector over the Fast-Hessian detector (F:abst/feature/detect/interest/WrapFHtoInterestPoint.java:36-90).  On the GPU it
    only runs fused with a describer (FactoryDetectDescribe.fuseTogether); stand-alone detection is FastHessianFeatureDetector."""

    def __init__(self, config=None):
        self.config = config or ConfigFastHessian()


class FactoryInterestPoint:
    @staticmethod
    def fastHessian(config=None):
        """F:factory/feature/detect/interest/FactoryInterestPoint.java:127-130"""
        return WrapFHtoInterestPoint(config)


class DetectDescribeFusion(DetectDescribePoint):
    """DetectDescribePoint<T,TupleDesc_B> = DetectDescribeFusion(fastHessian, null, brief) (F:abst/feature/detdesc/DetectDescribeFusion.java:
    45-165): Fast-Hessian points in detector order, every one described (WrapDescribeBrief.process always returns true), orientation 0
    (WrapFHtoInterestPoint.getOrientation), radius = scale * 2.  Detection, BRIEF and -- through associateImages -- Hamming association run
    on the device without the points or words leaving it (bhip_surf_create_brief)."""

    def __init__(self, detector, describe, ctx=None):
        self.ctx = ctx or Context.default()
        L = _lib.load()
        fh = detector.config._c()
        d = describe.definition
        self.describe = describe
        self._words = (describe.length + 31) // 32
        h = C.c_void_p()
        _check(self.ctx, L.bhip_surf_create_brief(self.ctx._h, C.byref(fh), d.radius, describe.length, d.samplePoints.ctypes.data_as(_lib._i32p),
                                                  d.compare.ctypes.data_as(_lib._i32p), C.byref(h)))
        self._h = h
        self.ctx._children.add(self)
        self._dof = 0
        self._batch = 0
        self._image = 0
        self._cache = {}

    def createDescription(self):
        return TupleDesc_B(self.describe.length)

    def getDescriptionType(self):
        return TupleDesc_B

    def _results(self, image=None):
        image = self._image if image is None else image
        if image not in self._cache:
            L = _lib.load()
            n = C.c_int(0)
            _check(self.ctx, L.bhip_surf_count(self._h, image, C.byref(n)))
            n = n.value
            xys = np.zeros((n, 3)); ang = np.zeros(n); white = np.zeros(n, dtype=np.uint8); words = np.zeros((n, self._words), dtype=np.int32)
            if n:
                _check(self.ctx, L.bhip_surf_fetch(self._h, image, xys.ctypes.data_as(_lib._dp), ang.ctypes.data_as(_lib._dp),
                                                   white.ctypes.data_as(_lib._u8p), None))
                _check(self.ctx, L.bhip_surf_fetch_brief(self._h, image, words.ctypes.data_as(_lib._i32p)))
            self._cache[image] = (xys, ang, white, words)
        return self._cache[image]

    def fetchAll(self, out=None):
        """(xy_scale [total,3], words [total, ceil(numPoints/32)] int32, starts [batch+1]) of the whole last batch."""
        L = _lib.load()
        counts = self.counts()
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        total = int(starts[-1])
        xys = np.empty((total, 3)); words = np.empty((total, self._words), dtype=np.int32)
        if total:
            _check(self.ctx, L.bhip_surf_fetch_all(self._h, xys.ctypes.data_as(_lib._dp), None, None, None))
            _check(self.ctx, L.bhip_surf_fetch_brief(self._h, -1, words.ctypes.data_as(_lib._i32p)))
        return xys, words, starts

    def associateImages(self, srcImages, dstImages, maxError=Double_MAX_VALUE, backwardsValidation=True):
        """Greedy Hamming association (ScoreAssociateHamming_B) of image srcImages[p] with image dstImages[p] of the last detect, on the
        words still resident on the device (bhip_assoc_hamming_surf)."""
        L = _lib.load()
        a = np.ascontiguousarray(srcImages, dtype=np.int32)
        b = np.ascontiguousarray(dstImages, dtype=np.int32)
        if a.shape != b.shape:
            raise IllegalArgumentException("source and destination image lists differ in length")
        total = max(self.totalFeatures(), 1)
        pairs = np.full(total, -1, dtype=np.int32)
        fit = np.zeros(total)
        _check(self.ctx, L.bhip_assoc_hamming_surf(self._h, len(a), a.ctypes.data_as(_lib._ip), b.ctypes.data_as(_lib._ip), float(maxError),
                                                   1 if backwardsValidation else 0, pairs.ctypes.data_as(_lib._ip), fit.ctypes.data_as(_lib._dp)))
        return pairs, fit

    def deviceViewBrief(self, image):
        """(dev_words_ptr, ints_per_feature, n) of image `image` -- valid until the next detect."""
        d, w, n = C.c_void_p(), C.c_int(0), C.c_int(0)
        _check(self.ctx, _lib.load().bhip_surf_dev_view_brief(self._h, image, C.byref(d), C.byref(w), C.byref(n)))
        return d.value, w.value, n.value

    def describePoints(self, xy_scale, image=0):
        raise RuntimeError("a fused BRIEF object describes the points it detects; use DescribePointBrief for a caller's point list")

    def getOrientation(self, featureIndex):
        return 0.0

    def getDescription(self, index):
        return TupleDesc_B(self.describe.length, self._results()[3][index])

    def hasOrientation(self):
        return False   # orientation == null -> detector.hasOrientation() (DetectDescribeFusion.java:150-155)


class FactoryDetectDescribe:
    @staticmethod
    def fuseTogether(detector, orientation, describe, ctx=None):
        """F:factory/feature/detdesc/FactoryDetectDescribe.java:279-284.  On the GPU: Fast-Hessian + (no orientation) + fixed BRIEF."""
        if not isinstance(detector, WrapFHtoInterestPoint) or orientation is not None or not isinstance(describe, WrapDescribeBrief):
            raise RuntimeError("only fuseTogether(fastHessian, null, brief) is implemented on the GPU (use the Java path)")
        return DetectDescribeFusion(detector, describe, ctx=ctx)

    @staticmethod
    def surfColorFast(configDetector=None, configDesc=None, configOrientation=None, imageType=None, ctx=None):
        """F:factory/feature/detdesc/FactoryDetectDescribe.java:154-176"""
        if not isinstance(imageType, PlanarType):
            raise IllegalArgumentException("Image type not supported")
        return SurfPlanar_to_DetectDescribePoint(False, configDetector, configDesc, configOrientation, imageType.numBands, ctx)

    @staticmethod
    def surfColorStable(configDetector=None, configDescribe=None, configOrientation=None, imageType=None, ctx=None):
        """F:factory/feature/detdesc/FactoryDetectDescribe.java:246-268"""
        if not isinstance(imageType, PlanarType):
            raise IllegalArgumentException("Image type not supported")
        return SurfPlanar_to_DetectDescribePoint(True, configDetector, configDescribe, configOrientation, imageType.numBands, ctx)

    @staticmethod
    def surfFast(configDetector=None, configDesc=None, configOrientation=None, imageType=GrayF32, ctx=None):
        if imageType is not GrayF32 and imageType is not GrayU8:
            raise RuntimeError("only GrayF32 and GrayU8 are implemented on the GPU (use the Java path)")
        return DetectDescribePoint(False, configDetector, configDesc, configOrientation, ctx)

    @staticmethod
    def surfStable(configDetector=None, configDescribe=None, configOrientation=None, imageType=GrayF32, ctx=None):
        if imageType is not GrayF32 and imageType is not GrayU8:
            raise RuntimeError("only GrayF32 and GrayU8 are implemented on the GPU (use the Java path)")
        return DetectDescribePoint(True, configDetector, configDescribe, configOrientation, ctx)


# ------------------------------------------------------------------------------------------------------------------
# association
# ------------------------------------------------------------------------------------------------------------------
class ScoreAssociateEuclideanSq_F64:
    """F:abst/feature/associate/ScoreAssociateEuclideanSq_F64.java -> DescriptorDistance.euclideanSq"""
    kind, sqrt = "l2", 0

    def getScoreType(self):
        return MatchScoreType.NORM_ERROR


class ScoreAssociateEuclidean_F64:
    kind, sqrt = "l2", 1

    def getScoreType(self):
        return MatchScoreType.NORM_ERROR


class ScoreAssociateHamming_B:
    kind, sqrt = "hamming", 0

    def getScoreType(self):
        return MatchScoreType.NORM_ERROR


class AssociateDescription:
    """WrapAssociateGreedy (F:abst/feature/associate/WrapAssociateGreedy.java:73-123) over AssociateGreedy on the GPU."""

    def __init__(self, score, maxError, backwardsValidation, ctx=None):
        self.ctx = ctx or Context.default()
        self.score = score
        self.maxFitError = float(maxError)
        self.backwardsValidation = bool(backwardsValidation)
        self.listSrc = None
        self.listDst = None
        self._matches = []
        self._unassocSrc = []
        self._pairs = np.zeros(0, dtype=np.int32)
        self._fit = np.zeros(0)
        self._nd = 0

    def setSource(self, listSrc):
        self.listSrc = listSrc

    def setDestination(self, listDst):
        self.listDst = listDst

    @staticmethod
    def _pack(lst, kind):
        if isinstance(lst, np.ndarray):
            return np.ascontiguousarray(lst, dtype=np.float64 if kind == "l2" else np.int32)
        if len(lst) == 0:
            return np.zeros((0, 1), dtype=np.float64 if kind == "l2" else np.int32)
        if kind == "l2":
            return np.ascontiguousarray(np.stack([np.asarray(d.value, dtype=np.float64) for d in lst]))
        return np.ascontiguousarray(np.stack([np.asarray(d.data, dtype=np.int32) for d in lst]))

    def associate(self):
        if self.listSrc is None:
            raise IllegalArgumentException("source features not specified")
        if self.listDst is None:
            raise IllegalArgumentException("destination features not specified")
        kind = self.score.kind
        src = self._pack(self.listSrc, kind)
        dst = self._pack(self.listDst, kind)
        ns, nd = len(src), len(dst)
        length = src.shape[1] if ns else (dst.shape[1] if nd else 1)
        if ns and nd and src.shape[1] != dst.shape[1]:
            raise IllegalArgumentException("descriptor lengths differ")
        pairs, fit = _PinnedPool.arrays(self.ctx, [((ns,), np.int32), ((ns,), np.float64)])
        pairs[:] = -1
        fit[:] = self.maxFitError
        L = _lib.load()
        if ns:
            if kind == "l2":
                _check(self.ctx, L.bhip_assoc_l2_f64(self.ctx._h, src.ctypes.data_as(_lib._dp), ns, dst.ctypes.data_as(_lib._dp), nd, length,
                                                     self.maxFitError, int(self.backwardsValidation), self.score.sqrt,
                                                     pairs.ctypes.data_as(_lib._ip), fit.ctypes.data_as(_lib._dp)))
            else:
                _check(self.ctx, L.bhip_assoc_hamming(self.ctx._h, src.ctypes.data_as(_lib._i32p), ns, dst.ctypes.data_as(_lib._i32p), nd, length,
                                                      self.maxFitError, int(self.backwardsValidation), pairs.ctypes.data_as(_lib._ip),
                                                      fit.ctypes.data_as(_lib._dp)))
        self._pairs, self._fit, self._nd = pairs, fit, nd
        self._matches = None      # the object lists are built when they are asked for (getMatches / getUnassociatedSource)
        self._unassocSrc = None

    @property
    def matches(self):
        if self._matches is None:
            idx = np.nonzero(self._pairs >= 0)[0]
            self._matches = [AssociatedIndex(i, d, f) for i, d, f in zip(idx.tolist(), self._pairs[idx].tolist(), self._fit[idx].tolist())]
        return self._matches

    @property
    def unassocSrc(self):
        if self._unassocSrc is None:
            self._unassocSrc = np.nonzero(self._pairs < 0)[0].tolist()
        return self._unassocSrc

    def getPairs(self):
        return self._pairs

    def getFitQuality(self):
        return self._fit

    def getMatches(self):
        return self.matches

    def getUnassociatedSource(self):
        return self.unassocSrc

    def getUnassociatedDestination(self):
        # FindUnassociated.checkDestination (F:alg/feature/associate/FindUnassociated.java:56-72)
        matched = np.zeros(self._nd, dtype=bool)
        for m in self.matches:
            matched[m.dst] = True
        return [i for i in range(self._nd) if not matched[i]]

    def setMaxScoreThreshold(self, score):
        self.maxFitError = float(score)

    def getScoreType(self):
        return self.score.getScoreType()

    def uniqueSource(self):
        return True

    def uniqueDestination(self):
        return self.backwardsValidation


class AssociateSurfBasic:
    """F:alg/feature/associate/AssociateSurfBasic.java:36-170: SURF features are split by the sign of the Laplacian (BrightFeature.white)
    and each sign is associated on its own -- features of different sign are never matched and each contraction is a quarter of the size."""

    def __init__(self, assoc):
        self.assoc = assoc
        self._src = ([], [])   # (positive, negative) lists of (index, feature)
        self._dst = ([], [])
        self.matches = []
        self.unassociatedSrc = []

    @staticmethod
    def _sort(features):
        pos, neg = [], []
        for i, f in enumerate(features):
            (pos if f.white else neg).append((i, f))
        return pos, neg

    def setSrc(self, src):
        self._src = self._sort(src)

    def setDst(self, dst):
        self._dst = self._sort(dst)

    def swapLists(self):
        self._src, self._dst = self._dst, self._src

    def associate(self):
        self.matches = []
        self.unassociatedSrc = []
        if not (self._src[0] or self._src[1]) or not (self._dst[0] or self._dst[1]):
            return
        for sign in (0, 1):   # positive, then negative
            s, d = self._src[sign], self._dst[sign]
            self.assoc.setSource([f for _, f in s])
            self.assoc.setDestination([f for _, f in d])
            self.assoc.associate()
            for a in self.assoc.getMatches():
                self.matches.append(AssociatedIndex(s[a.src][0], d[a.dst][0], a.fitScore))
            self.unassociatedSrc.extend(s[i][0] for i in self.assoc.getUnassociatedSource())

    def getMatches(self):
        return self.matches

    def totalDestination(self):
        return len(self._dst[0]) + len(self._dst[1])

    def getUnassociatedSrc(self):
        return self.unassociatedSrc

    def getAssoc(self):
        return self.assoc


class WrapAssociateSurfBasic:
    """F:abst/feature/associate/WrapAssociateSurfBasic.java:33-101"""

    def __init__(self, alg):
        self.alg = alg

    def setSource(self, listSrc):
        self.alg.setSrc(listSrc)

    def setDestination(self, listDst):
        self.alg.setDst(listDst)

    def associate(self):
        self.alg.associate()

    def getMatches(self):
        return self.alg.getMatches()

    def getUnassociatedSource(self):
        return self.alg.getUnassociatedSrc()

    def getUnassociatedDestination(self):
        # FindUnassociated.checkDestination (F:alg/feature/associate/FindUnassociated.java:56-72)
        n = self.alg.totalDestination()
        matched = np.zeros(n, dtype=bool)
        for m in self.alg.getMatches():
            matched[m.dst] = True
        return [i for i in range(n) if not matched[i]]

    def setMaxScoreThreshold(self, score):
        self.alg.getAssoc().setMaxScoreThreshold(score)

    def getScoreType(self):
        return self.alg.getAssoc().getScoreType()

    def uniqueSource(self):
        return self.alg.getAssoc().uniqueSource()

    def uniqueDestination(self):
        return self.alg.getAssoc().uniqueDestination()


class FactoryAssociation:
    @staticmethod
    def greedy(score, maxError, backwardsValidation, ctx=None):
        return AssociateDescription(score, maxError, backwardsValidation, ctx)

    @staticmethod
    def defaultScore(tupleType):
        """F:factory/feature/associate/FactoryAssociation.java:141-156"""
        if issubclass(tupleType, TupleDesc_F64):
            return ScoreAssociateEuclideanSq_F64()
        if tupleType is TupleDesc_B:
            return ScoreAssociateHamming_B()
        raise IllegalArgumentException("Unknown tuple type: %s" % tupleType)


# ------------------------------------------------------------------------------------------------------------------
# static op classes = what the BOverride* hooks replace
# ------------------------------------------------------------------------------------------------------------------
def _ctx(ctx):
    return ctx or Context.default()


class IntegralImageOps:
    @staticmethod
    def transform(input, transformed=None, ctx=None):
        """GIntegralImageOps.transform (I:alg/transform/ii/GIntegralImageOps.java:55-70)"""
        ctx = _ctx(ctx)
        if isinstance(input, GrayU8):
            # transform(GrayU8, GrayS32) (I:alg/transform/ii/impl/ImplIntegralImageOps.java:94-118)
            if transformed is None:
                transformed = GrayS32(input.width, input.height)
            elif not isinstance(transformed, GrayS32):
                raise IllegalArgumentException("GrayU8 is transformed into GrayS32")
            elif transformed.width != input.width or transformed.height != input.height:
                transformed.reshape(input.width, input.height)
            _check(ctx, _lib.load().bhip_integral_u8_s32(ctx._h, input._p(), input.startIndex, input.stride, input.width, input.height, transformed._p(),
                                                         transformed.startIndex, transformed.stride))
            return transformed
        if transformed is None:
            transformed = GrayF32(input.width, input.height)
        elif transformed.width != input.width or transformed.height != input.height:
            transformed.reshape(input.width, input.height)
        _check(ctx, _lib.load().bhip_integral_f32(ctx._h, input._p(), input.startIndex, input.stride, input.width, input.height, transformed._p(),
                                                  transformed.startIndex, transformed.stride))
        return transformed


class IntegralImageFeatureIntensity:
    @staticmethod
    def hessian(integral, skip, size, intensity, ctx=None):
        """F:alg/feature/detect/intensity/IntegralImageFeatureIntensity.java:43-56; intensity must be (width/skip) x (height/skip)."""
        ctx = _ctx(ctx)
        if isinstance(integral, GrayS32):
            _check(ctx, _lib.load().bhip_hessian_s32(ctx._h, integral._p(), integral.startIndex, integral.stride, integral.width, integral.height, skip, size,
                                                     intensity._p(), intensity.startIndex, intensity.stride))
            return
        _check(ctx, _lib.load().bhip_hessian_f32(ctx._h, integral._p(), integral.startIndex, integral.stride, integral.width, integral.height, skip, size,
                                                 intensity._p(), intensity.startIndex, intensity.stride))


class NonMaxSuppression:
    """FactoryFeatureExtractor.nonmax(config) -> WrapperNonMaximumBlock(NonMaxBlock(NonMaxBlockSearchStrict.Max))
    (F:factory/feature/detect/extract/FactoryFeatureExtractor.java:63-102).  Only the strict, maxima-only extractor runs on the GPU;
    anything else raises RuntimeError, which is the BOverride convention for "use the Java code"."""

    def __init__(self, config, ctx=None):
        config = config or ConfigExtract()
        config.checkValidity()
        if not config.useStrictRule or config.detectMinimums or not config.detectMaximums:
            raise RuntimeError("only the strict maxima extractor is implemented on the GPU")
        self.ctx = _ctx(ctx)
        self.radius, self.threshold, self.border = config.radius, config.threshold, config.ignoreBorder

    def process(self, intensity, candidateMin=None, candidateMax=None, foundMin=None, foundMax=None):
        cap = max(1, ((intensity.width + self.radius) // (self.radius + 1)) * ((intensity.height + self.radius) // (self.radius + 1)))
        xy = np.zeros((cap, 2), dtype=np.int16)
        n = C.c_int(0)
        _check(self.ctx, _lib.load().bhip_nonmax_block_f32(self.ctx._h, intensity._p(), intensity.startIndex, intensity.stride, intensity.width,
                                                          intensity.height, self.radius, self.threshold, self.border, xy.ctypes.data_as(_lib._i16p), cap,
                                                          C.byref(n)))
        out = [Point2D_I16(int(x), int(y)) for x, y in xy[:n.value]]
        if foundMax is not None:
            del foundMax[:]
            foundMax.extend(out)
        return out

    def getSearchRadius(self): return self.radius
    def setSearchRadius(self, r): self.radius = r
    def getIgnoreBorder(self): return self.border
    def setIgnoreBorder(self, b): self.border = b
    def getThresholdMaximum(self): return self.threshold
    def setThresholdMaximum(self, t): self.threshold = t
    def getUsesCandidates(self): return False
    def canDetectMaximums(self): return True
    def canDetectMinimums(self): return False


class FactoryFeatureExtractor:
    @staticmethod
    def nonmax(config=None, ctx=None):
        return NonMaxSuppression(config, ctx)


class SelectNBestFeatures:
    """F:alg/feature/detect/extract/SelectNBestFeatures.java:31-97: keep the N most intense corners.  The order of the kept corners is
    the order ddogleg's QuickSelect leaves them in; the GPU runs the restated routine (see include/boofhip.h: order unpinned vs the jar)."""

    def __init__(self, N, ctx=None):
        self.ctx = _ctx(ctx)
        self.target = N
        self.bestCorners = []

    def setN(self, N):
        self.target = N

    def process(self, intensityImage, origCorners, positive):
        xy = np.array([[p.x, p.y] for p in origCorners], dtype=np.int16).reshape(-1, 2)
        out = np.zeros((max(len(xy), 1), 2), dtype=np.int16)
        n = C.c_int(0)
        _check(self.ctx, _lib.load().bhip_select_nbest_f32(self.ctx._h, intensityImage._p(), intensityImage.startIndex, intensityImage.stride,
                                                          intensityImage.width, intensityImage.height, xy.ctypes.data_as(_lib._i16p), len(xy),
                                                          int(self.target), 1 if positive else 0, out.ctypes.data_as(_lib._i16p), C.byref(n)))
        self.bestCorners = [Point2D_I16(int(x), int(y)) for x, y in out[:n.value]]

    def getBestCorners(self):
        return self.bestCorners


class GradientCornerIntensity:
    """FactoryIntensityPointAlg.shiTomasi / harris (unweighted, GrayF32 derivatives): ImplSsdCorner_F32 with ShiTomasiCorner_F32 /
    HarrisCorner_F32 (F:alg/feature/detect/intensity/impl/ImplSsdCorner_F32.java:62-196).  Single-threaded summation order."""

    def __init__(self, kind, windowRadius, kappa=0.0, ctx=None):
        self.kind, self.radius, self.kappa = kind, int(windowRadius), float(kappa)
        self.ctx = _ctx(ctx)

    def getRadius(self):
        return self.radius

    def getIgnoreBorder(self):
        return self.radius

    def process(self, derivX, derivY, intensity):
        if derivX.width != derivY.width or derivX.height != derivY.height:
            raise IllegalArgumentException("Image shapes do not match")   # InputSanityCheck.checkSameShape
        if (derivX.startIndex, derivX.stride) != (derivY.startIndex, derivY.stride):
            raise IllegalArgumentException("derivX and derivY must share startIndex and stride")
        intensity.reshape(derivX.width, derivX.height)
        _check(self.ctx, _lib.load().bhip_corner_intensity_f32(self.ctx._h, self.kind, self.radius, self.kappa, derivX._p(), derivY._p(), derivX.startIndex,
                                                               derivX.stride, derivX.width, derivX.height, intensity._p(), intensity.startIndex,
                                                               intensity.stride))


class FactoryIntensityPointAlg:
    @staticmethod
    def shiTomasi(windowRadius, weighted=False, derivType=None, ctx=None):
        """F:factory/feature/detect/intensity/FactoryIntensityPointAlg.java:132-160"""
        if weighted or (derivType is not None and derivType is not GrayF32):
            raise RuntimeError("only the unweighted GrayF32 corner intensity is implemented on the GPU (use the Java path)")
        return GradientCornerIntensity(0, windowRadius, 0.0, ctx)

    @staticmethod
    def harris(windowRadius, kappa, weighted=False, derivType=None, ctx=None):
        """F:factory/feature/detect/intensity/FactoryIntensityPointAlg.java:91-118"""
        if weighted or (derivType is not None and derivType is not GrayF32):
            raise RuntimeError("only the unweighted GrayF32 corner intensity is implemented on the GPU (use the Java path)")
        return GradientCornerIntensity(1, windowRadius, kappa, ctx)


class GeneralFeatureDetector:
    """F:alg/feature/detect/interest/GeneralFeatureDetector.java:67-160 for a gradient corner intensity and a maxima extractor:
    intensity.process -> extractor.process -> selectBest (maxFeatures > 0: SelectNBestFeatures, :143-160).  Exclusion lists are not
    mirrored (the trackers that pass them stay in Java)."""

    def __init__(self, intensity, extractor):
        self.intensity, self.extractor = intensity, extractor
        if intensity.getIgnoreBorder() > extractor.getIgnoreBorder():
            extractor.setIgnoreBorder(intensity.getIgnoreBorder())
        self.maxFeatures = 0
        self.intensityImage = GrayF32(1, 1)
        self.foundMaximum = []
        self.selectBest = SelectNBestFeatures(10, intensity.ctx)

    def setMaxFeatures(self, n):
        self.maxFeatures = n

    def getRequiresGradient(self):
        return True

    def getRequiresHessian(self):
        return False

    def setThreshold(self, threshold):
        self.extractor.setThresholdMaximum(threshold)

    def getThreshold(self):
        return self.extractor.getThresholdMaximum()

    def process(self, image, derivX, derivY, derivXX=None, derivYY=None, derivXY=None):
        self.intensity.process(derivX, derivY, self.intensityImage)
        self.foundMaximum = self.extractor.process(self.intensityImage)
        if self.maxFeatures > 0:   # GeneralFeatureDetector.java:143-160 (numSelectMax = maxFeatures without an exclusion list)
            self.selectBest.setN(self.maxFeatures)
            self.selectBest.process(self.intensityImage, self.foundMaximum, True)
            self.foundMaximum = list(self.selectBest.getBestCorners())

    def getIntensity(self):
        return self.intensityImage

    def getMaximums(self):
        return self.foundMaximum


class FastHessianFeatureDetector:
    """FactoryInterestPointAlgs.fastHessian(config).detect(integral) (F:alg/feature/detect/interest/FastHessianFeatureDetector.java:156-188)"""

    def __init__(self, config=None, ctx=None):
        self.config = config or ConfigFastHessian()
        self.ctx = _ctx(ctx)
        self.foundPoints = np.zeros((0, 3))

    def detect(self, integral):
        cfg = self.config._c()
        cap = 1 << 15
        while True:
            out = np.zeros((cap, 3))
            n = C.c_int(0)
            fn = _lib.load().bhip_fh_detect_s32 if isinstance(integral, GrayS32) else _lib.load().bhip_fh_detect_f32
            _check(self.ctx, fn(self.ctx._h, C.byref(cfg), integral._p(), integral.startIndex, integral.stride, integral.width,
                                                           integral.height, out.ctypes.data_as(_lib._dp), cap, C.byref(n)))
            if n.value <= cap:
                self.foundPoints = out[:n.value].copy()
                return
            cap = n.value

    def getFoundPoints(self):
        return self.foundPoints


@dataclass
class Kernel1D_F32:
    """T:struct/convolve/Kernel1D_F32.java: data, width, offset (origin index; width/2 by default)"""
    data: np.ndarray
    width: int = 0
    offset: int = -1

    def __post_init__(self):
        self.data = np.ascontiguousarray(self.data, dtype=np.float32)
        self.width = len(self.data)
        if self.offset < 0:
            self.offset = self.width // 2


def _conv(fn, kernel, src, dst, ctx):
    ctx = _ctx(ctx)
    for im in (src, dst):
        _check_extent(im.width, im.height, im.startIndex, im.stride, im.data.size)
    if dst.width != src.width or dst.height != src.height:
        raise IllegalArgumentException("Image shapes do not match")  # InputSanityCheck.checkSameShape
    _check(ctx, fn(ctx._h, kernel.data.ctypes.data_as(_lib._fp), kernel.width, kernel.offset, src._p(), src.startIndex, src.stride, src.width, src.height,
                   dst._p(), dst.startIndex, dst.stride))


class ConvolveImageNoBorder:
    """BOverrideConvolveImage.horizontal/vertical/convolve targets (I:alg/filter/convolve/ConvolveImageNoBorder.java:53-90)"""

    @staticmethod
    def convolve(kernel, input, output, ctx=None):
        ctx = _ctx(ctx)
        if output.width != input.width or output.height != input.height:
            raise IllegalArgumentException("Image shapes do not match")
        _check(ctx, _lib.load().bhip_conv2d_f32(ctx._h, kernel.data.ctypes.data_as(_lib._fp), kernel.width, kernel.offset, input._p(), input.startIndex,
                                                input.stride, input.width, input.height, output._p(), output.startIndex, output.stride))

    @staticmethod
    def horizontal(kernel, input, output, ctx=None):
        _conv(_lib.load().bhip_conv_h_f32, kernel, input, output, ctx)

    @staticmethod
    def vertical(kernel, input, output, ctx=None):
        _conv(_lib.load().bhip_conv_v_f32, kernel, input, output, ctx)


class ConvolveImageNormalized:
    """BOverrideConvolveImageNormalized targets (I:alg/filter/convolve/ConvolveImageNormalized.java:48-93)"""

    @staticmethod
    def horizontal(kernel, src, dst, ctx=None):
        _conv(_lib.load().bhip_conv_norm_h_f32, kernel, src, dst, ctx)

    @staticmethod
    def vertical(kernel, src, dst, ctx=None):
        _conv(_lib.load().bhip_conv_norm_v_f32, kernel, src, dst, ctx)


class FactoryKernelGaussian:
    @staticmethod
    def gaussian1D_F32(sigma, radius):
        """FactoryKernelGaussian.gaussian(Kernel1D_F32.class, sigma, radius) (I:factory/filter/kernel/FactoryKernelGaussian.java:120-153)"""
        if sigma <= 0 and radius <= 0:
            raise IllegalArgumentException("Sigma must be > 0")
        L = _lib.load()
        w = -L.bhip_gaussian_kernel1d_f32(float(sigma), int(radius), None, 0)
        out = np.zeros(w, dtype=np.float32)
        L.bhip_gaussian_kernel1d_f32(float(sigma), int(radius), out.ctypes.data_as(_lib._fp), w)
        return Kernel1D_F32(out)


class ConvolveImageDownNormalized:
    """I:alg/filter/convolve/ConvolveImageDownNormalized.java:53-86 (the NORMALIZED ConvolveDown of the discrete pyramid)"""

    @staticmethod
    def _run(fn, kernel, image, dest, skip, ctx):
        ctx = _ctx(ctx)
        rc = fn(ctx._h, kernel.data.ctypes.data_as(_lib._fp), kernel.width, image._p(), image.startIndex, image.stride, image.width, image.height,
                dest._p(), dest.startIndex, dest.stride, dest.width, dest.height, int(skip))
        _check(ctx, rc)

    @staticmethod
    def horizontal(kernel, image, dest, skip, ctx=None):
        ConvolveImageDownNormalized._run(_lib.load().bhip_conv_down_norm_h_f32, kernel, image, dest, skip, ctx)

    @staticmethod
    def vertical(kernel, image, dest, skip, ctx=None):
        ConvolveImageDownNormalized._run(_lib.load().bhip_conv_down_norm_v_f32, kernel, image, dest, skip, ctx)


class PyramidDiscreteSampleBlur:
    """I:alg/transform/pyramid/PyramidDiscreteSampleBlur.java:48-126: layer i = layer i-1 blurred with the (border-normalised) kernel and
    sub-sampled by scale[i]/scale[i-1]; layer 0 = the input when scale[0] == 1."""

    def __init__(self, kernel, sigma, saveOriginalReference, scaleFactors, ctx=None):
        self.ctx = _ctx(ctx)
        self.kernel = kernel
        self.saveOriginalReference = bool(saveOriginalReference)
        self.scale = [int(s) for s in scaleFactors]
        # ImagePyramidBase.checkScales (T:struct/pyramid/ImagePyramidBase.java:100-112)
        if self.scale[0] < 0:
            raise IllegalArgumentException("The first layer must be more than zero.")
        for a, b in zip(self.scale, self.scale[1:]):
            if b < a:
                raise IllegalArgumentException("Higher layers must be the same size or larger than previous layers.")
        self.sigmas = [0.0] * len(self.scale)
        for i in range(1, len(self.scale)):
            prev, applied = self.sigmas[i - 1], sigma * self.scale[i - 1]
            self.sigmas[i] = math.sqrt(prev * prev + applied * applied)
        self.layers = None

    def getNumLayers(self):
        return len(self.scale)

    def getScale(self, layer):
        return float(self.scale[layer])

    def getSigma(self, layer):
        return self.sigmas[layer]

    def getSampleOffset(self, layer):
        return 0.0

    def process(self, input):
        L = _lib.load()
        n = len(self.scale)
        sc = np.asarray(self.scale, dtype=np.int32)
        dims = np.zeros(2 * n, dtype=np.int32)
        offs = np.zeros(n, dtype=np.int64)
        total = C.c_longlong(0)
        if L.bhip_pyramid_layout(input.width, input.height, sc.ctypes.data_as(_lib._ip), n, dims.ctypes.data_as(_lib._ip),
                                 offs.ctypes.data_as(_lib._llp), C.byref(total)) != 0:
            raise IllegalArgumentException("bad pyramid scales")
        packed = np.zeros(total.value, dtype=np.float32)
        rc = L.bhip_pyramid_f32(self.ctx._h, self.kernel.data.ctypes.data_as(_lib._fp), self.kernel.width, sc.ctypes.data_as(_lib._ip), n, input._p(),
                                input.startIndex, input.stride, input.width, input.height, packed.ctypes.data_as(_lib._fp))
        _check(self.ctx, rc)
        self.layers = []
        for i in range(n):
            w, h = int(dims[2 * i]), int(dims[2 * i + 1])
            if i == 0 and self.scale[0] == 1 and self.saveOriginalReference:
                self.layers.append(input)  # setFirstLayer(input)
            else:
                self.layers.append(GrayF32(w, h, packed[offs[i]:offs[i] + w * h]))
        return self

    def getLayer(self, i):
        return self.layers[i]

    def getWidth(self, i):
        return self.layers[i].width

    def getHeight(self, i):
        return self.layers[i].height


class FactoryPyramid:
    @staticmethod
    def discreteGaussian(scaleFactors, sigma, radius, saveOriginalReference=False, ctx=None):
        """I:factory/transform/pyramid/FactoryPyramid.java:53-61"""
        kernel = FactoryKernelGaussian.gaussian1D_F32(sigma, radius)
        return PyramidDiscreteSampleBlur(kernel, sigma, saveOriginalReference, scaleFactors, ctx)


@dataclass
class Kernel2D_F32:
    """T:struct/convolve/Kernel2D_F32.java: width x width values row-major, offset = origin index along both axes"""
    data: np.ndarray
    width: int = 0
    offset: int = -1

    def __post_init__(self):
        self.data = np.ascontiguousarray(self.data, dtype=np.float32)
        self.width = self.data.shape[0]
        if self.data.ndim != 2 or self.data.shape[1] != self.width:
            raise IllegalArgumentException("square kernel expected")
        if self.offset < 0:
            self.offset = self.width // 2


class BlurImageOps:
    @staticmethod
    def mean(input, output, radiusX, radiusY=None, storage=None, ctx=None):
        """BOverrideBlurImageOps.mean target (I:alg/filter/blur/BlurImageOps.java:343-376)"""
        ctx = _ctx(ctx)
        radiusY = radiusX if radiusY is None else radiusY
        if radiusX <= 0 or radiusY <= 0:
            raise IllegalArgumentException("Radius must be > 0")
        if output is None:
            output = GrayF32(input.width, input.height)
        _check(ctx, _lib.load().bhip_mean_f32(ctx._h, input._p(), input.startIndex, input.stride, input.width, input.height, int(radiusX), int(radiusY),
                                              output._p(), output.startIndex, output.stride))
        return output

    @staticmethod
    def median(input, output, radius, ctx=None):
        """BOverrideBlurImageOps.median target (I:alg/filter/blur/BlurImageOps.java:752-765)"""
        ctx = _ctx(ctx)
        if radius <= 0:
            raise IllegalArgumentException("Radius must be > 0")
        if output is None:
            output = GrayF32(input.width, input.height)
        _check(ctx, _lib.load().bhip_median_f32(ctx._h, input._p(), input.startIndex, input.stride, input.width, input.height, int(radius), output._p(),
                                                output.startIndex, output.stride))
        return output

    @staticmethod
    def gaussian(input, output, sigma, radius, storage=None, ctx=None):
        """BOverrideBlurImageOps.gaussian target (I:alg/filter/blur/BlurImageOps.java:406-425)"""
        ctx = _ctx(ctx)
        if output is None:
            output = GrayF32(input.width, input.height)
        _check(ctx, _lib.load().bhip_gaussian_f32(ctx._h, input._p(), input.startIndex, input.stride, input.width, input.height, float(sigma), int(radius),
                                                  output._p(), output.startIndex, output.stride))
        return output


class _Gradient:
    fn = None

    @classmethod
    def process(cls, orig, derivX, derivY, border=None, ctx=None):
        """border: None = null (frame untouched) or 0 = ImageBorderValue(0)"""
        ctx = _ctx(ctx)
        if border not in (None, 0):
            raise RuntimeError("border policy not implemented on the GPU")
        _check(ctx, getattr(_lib.load(), cls.fn)(ctx._h, orig._p(), orig.startIndex, orig.stride, orig.width, orig.height, derivX._p(), derivY._p(),
                                                 derivX.startIndex, derivX.stride, 0 if border is None else 1))


class GradientSobel(_Gradient):
    """I:alg/filter/derivative/GradientSobel.java:158-173"""
    fn = "bhip_sobel_f32"


class GradientThree(_Gradient):
    """I:alg/filter/derivative/GradientThree.java -> impl/GradientThree_Standard.java:40-62"""
    fn = "bhip_three_f32"


class DescribePointBrief:
    """DescribePointBrief.process for a list of points (F:alg/feature/describe/DescribePointBrief.java:73-89).  As in this fork of the
    reference, the fixed BRIEF variant samples the UNblurred image (SURVEY finding 6)."""

    def __init__(self, radius, samplePoints, compare, ctx=None):
        self.ctx = _ctx(ctx)
        self.radius = int(radius)
        self.samplePoints = np.ascontiguousarray(samplePoints, dtype=np.int32)
        self.compare = np.ascontiguousarray(compare, dtype=np.int32)
        self.image = None

    def setImage(self, image):
        self.image = image

    def processAll(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        n, npts = len(xy), len(self.compare)
        out = np.zeros((n, (npts + 31) // 32), dtype=np.int32)
        im = self.image
        if isinstance(im, GrayU8):   # ImplDescribeBinaryCompare_U8
            _check(self.ctx, _lib.load().bhip_brief_u8(self.ctx._h, im._p(), im.startIndex, im.stride, im.width, im.height, self.radius, npts,
                                                      self.samplePoints.ctypes.data_as(_lib._i32p), self.compare.ctypes.data_as(_lib._i32p),
                                                      xy.ctypes.data_as(_lib._dp), n, out.ctypes.data_as(_lib._i32p)))
            return out
        _check(self.ctx, _lib.load().bhip_brief_f32(self.ctx._h, im._p(), im.startIndex, im.stride, im.width, im.height, self.radius, npts,
                                                   self.samplePoints.ctypes.data_as(_lib._i32p), self.compare.ctypes.data_as(_lib._i32p),
                                                   xy.ctypes.data_as(_lib._dp), n, out.ctypes.data_as(_lib._i32p)))
        return out

    def process(self, c_x, c_y, feature):
        feature.data[:] = self.processAll([[c_x, c_y]])[0]
