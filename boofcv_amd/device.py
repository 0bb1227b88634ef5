"""Device-resident, batched forms of the boofcv-ip front end (bhip_*_dev_f32 of include/boofhip.h) on torch CUDA tensors.

torch is used for what it is good at here -- device memory and the current stream; every operation is one call into libboofhip.so
(hand-written HIP kernels).  A batch is a [B, H, W] float32 tensor whose last dimension is contiguous; rows may be strided (views work).
Names follow the reference classes the host-buffer forms in boofcv_amd/api.py mirror:

  DeviceImageOps.convolve*      ConvolveImageNoBorder / ConvolveImageNormalized      I:alg/filter/convolve/*.java
  DeviceImageOps.gaussian       BlurImageOps.gaussian                               I:alg/filter/blur/BlurImageOps.java:406-425
  DeviceImageOps.sobel / three  GradientSobel / GradientThree .process              I:alg/filter/derivative/GradientSobel.java:158-173
  DeviceImageOps.intensity      GradientToEdgeFeatures.intensityE / intensityAbs    F:alg/feature/detect/edge/GradientToEdgeFeatures.java:61-95
  DeviceImageOps.nonmax         NonMaxBlock.process (strict)                        F:alg/feature/detect/extract/NonMaxBlock.java:69-94
  DeviceImageOps.pyramid        PyramidDiscreteSampleBlur.process                   I:alg/transform/pyramid/PyramidDiscreteSampleBlur.java:88-118
  DeviceImageOps.cornerIntensity  GradientCornerIntensity.process                   F:alg/feature/detect/intensity/impl/ImplSsdCorner_F32.java:62-196
  DeviceImageOps.brief          DescribePointBrief.process                          F:alg/feature/describe/DescribePointBrief.java:73-89
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .api import Context, IllegalArgumentException, _check

INTENSITY_E, INTENSITY_ABS, INTENSITY_SQ = 0, 1, 2


def _geom(t):
    """(ptr, imageStride, rowStride, W, H, B) of a [B,H,W] float32 CUDA tensor with unit stride along x"""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or t.dtype != torch.float32 or not t.is_cuda:
        raise IllegalArgumentException("expected a [B,H,W] float32 CUDA tensor")
    B, H, W = t.shape
    if W > 1 and t.stride(2) != 1:
        raise IllegalArgumentException("the last dimension must be contiguous")
    return C.c_void_p(t.data_ptr()), (t.stride(0) if B > 1 else H * t.stride(1)), t.stride(1) if H > 1 else max(W, t.stride(1)), W, H, B


class DeviceImageOps:
    """All calls are asynchronous on the context's stream (create the Context on torch's current stream to mix with torch ops)."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or Context(device, stream=torch.cuda.current_stream(device).cuda_stream)
        self.L = _lib.load()
        self.device = torch.device("cuda", self.ctx.device)

    def _like(self, t):
        return torch.empty(t.shape, dtype=torch.float32, device=t.device)

    def _conv(self, fn, kernel, offset, src, out):
        k = np.ascontiguousarray(kernel, np.float32)
        out = self._like(src) if out is None else out
        ip, iis, irs, W, H, B = _geom(src)
        op, ois, ors, W2, H2, B2 = _geom(out)
        if (W, H, B) != (W2, H2, B2):
            raise IllegalArgumentException("input and output shapes differ")
        _check(self.ctx, fn(self.ctx._h, k.ctypes.data_as(_lib._fp), len(k), int(offset), ip, iis, irs, W, H, B, op, ois, ors))
        return out

    def convolveHorizontal(self, kernel, offset, src, out=None):
        return self._conv(self.L.bhip_conv_h_dev_f32, kernel, offset, src, out)

    def convolveVertical(self, kernel, offset, src, out=None):
        return self._conv(self.L.bhip_conv_v_dev_f32, kernel, offset, src, out)

    def convolveNormalizedHorizontal(self, kernel, offset, src, out=None):
        return self._conv(self.L.bhip_conv_norm_h_dev_f32, kernel, offset, src, out)

    def convolveNormalizedVertical(self, kernel, offset, src, out=None):
        return self._conv(self.L.bhip_conv_norm_v_dev_f32, kernel, offset, src, out)

    def gaussian(self, src, sigma, radius, out=None):
        out = self._like(src) if out is None else out
        ip, iis, irs, W, H, B = _geom(src)
        op, ois, ors, _, _, _ = _geom(out)
        _check(self.ctx, self.L.bhip_gaussian_dev_f32(self.ctx._h, ip, iis, irs, W, H, B, float(sigma), int(radius), op, ois, ors))
        return out

    def _grad(self, fn, src, border, dx, dy):
        """border: None = frame untouched (as the reference with a null border), 0 = ImageBorderValue(0)"""
        if dx is None:
            # with a border policy every pixel is written; without one the frame keeps what the caller put there (zeros here, filled on
            # torch's stream: make sure that fill is ordered before the kernel when the ctx runs on another stream)
            if border is None:
                dx, dy = torch.zeros(src.shape, dtype=torch.float32, device=src.device), torch.zeros(src.shape, dtype=torch.float32, device=src.device)
                torch.cuda.current_stream(src.device).synchronize()
            else:
                dx, dy = self._like(src), self._like(src)
        ip, iis, irs, W, H, B = _geom(src)
        xp, ois, ors, _, _, _ = _geom(dx)
        yp, ois2, ors2, _, _, _ = _geom(dy)
        if (ois, ors) != (ois2, ors2):
            raise IllegalArgumentException("derivX and derivY must share their layout")
        _check(self.ctx, fn(self.ctx._h, ip, iis, irs, W, H, B, xp, yp, ois, ors, 0 if border is None else 1))
        return dx, dy

    def sobel(self, src, border=0, dx=None, dy=None):
        return self._grad(self.L.bhip_sobel_dev_f32, src, border, dx, dy)

    def three(self, src, border=0, dx=None, dy=None):
        return self._grad(self.L.bhip_three_dev_f32, src, border, dx, dy)

    def intensity(self, kind, dx, dy, out=None):
        out = self._like(dx) if out is None else out
        xp, dis, drs, W, H, B = _geom(dx)
        yp, dis2, drs2, _, _, _ = _geom(dy)
        if (dis, drs) != (dis2, drs2):
            raise IllegalArgumentException("derivX and derivY must share their layout")
        op, ois, ors, _, _, _ = _geom(out)
        _check(self.ctx, self.L.bhip_gradient_intensity_dev_f32(self.ctx._h, int(kind), xp, yp, dis, drs, W, H, B, op, ois, ors))
        return out

    def nonmax(self, intensity, radius, threshold, border, cap=None):
        """-> (xy int16 [B, cap, 2], counts int32 [B]) on the device; lists are in the reference's block-raster order"""
        ip, iis, irs, W, H, B = _geom(intensity)
        if cap is None:
            step = radius + 1
            cap = max(1, ((max(W - 2 * border, 0) + step - 1) // step) * ((max(H - 2 * border, 0) + step - 1) // step))
        xy = torch.empty((B, cap, 2), dtype=torch.int16, device=intensity.device)
        n = torch.empty((B,), dtype=torch.int32, device=intensity.device)
        _check(self.ctx, self.L.bhip_nonmax_block_dev_f32(self.ctx._h, ip, iis, irs, W, H, B, int(radius), float(threshold), int(border),
                                                        C.c_void_p(xy.data_ptr()), cap, C.c_void_p(n.data_ptr())))
        return xy, n

    def cornerIntensity(self, kind, radius, kappa, dx, dy, out=None):
        """kind: 0 Shi-Tomasi, 1 Harris"""
        out = self._like(dx) if out is None else out
        xp, dis, drs, W, H, B = _geom(dx)
        yp, dis2, drs2, _, _, _ = _geom(dy)
        if (dis, drs) != (dis2, drs2):
            raise IllegalArgumentException("derivX and derivY must share their layout")
        op, ois, ors, _, _, _ = _geom(out)
        _check(self.ctx, self.L.bhip_corner_intensity_dev_f32(self.ctx._h, int(kind), int(radius), float(kappa), xp, yp, dis, drs, W, H, B, op, ois, ors))
        return out

    def pyramidLayout(self, width, height, scales):
        s = np.ascontiguousarray(scales, np.int32)
        dims = np.zeros(2 * len(s), np.int32)
        offs = np.zeros(len(s), np.int64)
        total = C.c_longlong()
        if self.L.bhip_pyramid_layout(width, height, s.ctypes.data_as(_lib._ip), len(s), dims.ctypes.data_as(_lib._ip), offs.ctypes.data_as(_lib._llp),
                                      C.byref(total)) != 0:
            raise IllegalArgumentException("bad pyramid scales")
        return dims.reshape(-1, 2), offs, total.value

    def pyramid(self, kernel, scales, src):
        """-> list of [B, h_i, w_i] layer views into one packed [B, total] tensor"""
        k = np.ascontiguousarray(kernel, np.float32)
        s = np.ascontiguousarray(scales, np.int32)
        ip, iis, irs, W, H, B = _geom(src)
        dims, offs, total = self.pyramidLayout(W, H, s)
        out = torch.empty((B, total), dtype=torch.float32, device=src.device)
        _check(self.ctx, self.L.bhip_pyramid_dev_f32(self.ctx._h, k.ctypes.data_as(_lib._fp), len(k), s.ctypes.data_as(_lib._ip), len(s), ip, iis, irs, W, H, B,
                                                   C.c_void_p(out.data_ptr())))
        return [out[:, int(offs[i]):int(offs[i]) + int(dims[i][0]) * int(dims[i][1])].view(B, int(dims[i][1]), int(dims[i][0])) for i in range(len(s))]

    def brief(self, img, radius, samplePoints, compare, xy, start):
        """xy: [N,2] float64 device tensor, start: host prefix (B+1) -> [N, ceil(numPoints/32)] int32 words"""
        ip, iis, irs, W, H, B = _geom(img)
        sp = np.ascontiguousarray(samplePoints, np.int32)
        cp = np.ascontiguousarray(compare, np.int32)
        st = np.ascontiguousarray(start, np.int32)
        if len(st) != B + 1:
            raise IllegalArgumentException("start must have batch+1 entries")
        words = (len(cp) + 31) // 32
        out = torch.empty((xy.shape[0], words), dtype=torch.int32, device=img.device)
        _check(self.ctx, self.L.bhip_brief_dev_f32(self.ctx._h, ip, iis, irs, W, H, B, int(radius), len(cp), sp.ctypes.data_as(_lib._i32p),
                                                 cp.ctypes.data_as(_lib._i32p), C.c_void_p(xy.data_ptr()), st.ctypes.data_as(_lib._ip),
                                                 C.c_void_p(out.data_ptr())))
        return out
