"""Restatement of OpenCV 4.2 ``cv::resize`` for float32 images (ORACLE; test infrastructure).

Third-party dependency of the reference, absent from /root/reference and from this
image: ``opencv-python==4.2.0.32`` (third_party_methods/environment.yaml:325).  Call
sites on the hot path:

  * INTER_LINEAR  input resize   third_party_methods/lib/datasets/data_augmentation_2d3d.py:510
  * INTER_CUBIC   peak patch x8  third_party_methods/lib/utils/paf_to_pose.py:123-124
  * INTER_CUBIC   PAF map   x8   third_party_methods/lib/utils/paf_to_pose.py:368-369
  * INTER_NEAREST (cpp variant)  third_party_methods/lib/utils/paf_to_pose.py:391-394

**Parity unpinned**: no fixture in the reference pins cv2's output and cv2 is not
installed here.  What is restated is OpenCV's published algorithm
(modules/imgproc/src/resize.cpp, 4.2.0):

  src coordinate   fx = (float)((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx
                   with scale_x = 1 / inv_scale_x held in double
  bilinear         taps (1-fx, fx); x index/weight clamp "sx<0 -> sx=0,fx=0",
                   "sx>=W-1 -> sx=W-1,fx=0"; rows clamped by index only
  bicubic          A = -0.75 (interpolateCubic), 4th coefficient = 1 - c0 - c1 - c2,
                   tap indices clamped to the image (replicate)
  evaluation order horizontal pass for each needed source row, then vertical pass,
                   all in float32, products summed left to right, NO fused multiply-add
                   (the scalar code path; OpenCV's SIMD paths may fuse -- one more
                   reason the boundary is unpinned)

The HIP kernels implement exactly this order, so HIP-vs-oracle is bit-exact even though
oracle-vs-real-cv2 can differ in the last ulp.
"""
import numpy as np

INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_CUBIC = 2

_f32 = np.float32


def cubic_coeffs(fx):
    """interpolateCubic (resize.cpp / imgproc precomp.hpp), float32, A = -0.75."""
    A = _f32(-0.75)
    x = _f32(fx)
    one = _f32(1.0)
    c0 = ((A * (x + one) - _f32(5) * A) * (x + one) + _f32(8) * A) * (x + one) - _f32(4) * A
    c1 = ((A + _f32(2)) * x - (A + _f32(3))) * x * x + one
    xm = one - x
    c2 = ((A + _f32(2)) * xm - (A + _f32(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def _src_coord(d, scale):
    """fx = (float)((d+0.5)*scale - 0.5); s = cvFloor(fx); fx -= s   (scale is a double)."""
    f = np.float32((d + 0.5) * scale - 0.5)
    s = int(np.floor(f))
    f = np.float32(f - np.float32(s))
    return s, f


def axis_tables(ssize, dsize, scale, interpolation):
    """Per-destination-index source offset and tap weights along one axis.

    Returns (ofs[dsize] int, coef[dsize, ktaps] float32).  For the x axis of INTER_LINEAR
    the index/weight clamp of resize.cpp is applied; ``ofs`` is NOT clamped for cubic or
    for the y axis (callers clamp tap indices, as OpenCV does).
    """
    k = {INTER_LINEAR: 2, INTER_CUBIC: 4}[interpolation]
    ofs = np.zeros(dsize, dtype=np.int64)
    coef = np.zeros((dsize, k), dtype=np.float32)
    for d in range(dsize):
        s, f = _src_coord(d, scale)
        ofs[d] = s
        if interpolation == INTER_CUBIC:
            coef[d] = cubic_coeffs(f)
        else:
            coef[d, 0] = _f32(1.0) - f
            coef[d, 1] = f
    return ofs, coef


def _linear_x_clamp(ofs, coef, ssize):
    ofs = ofs.copy()
    coef = coef.copy()
    for d in range(len(ofs)):
        if ofs[d] < 0:
            ofs[d] = 0
            coef[d] = (1.0, 0.0)
        if ofs[d] >= ssize - 1:
            ofs[d] = ssize - 1
            coef[d] = (1.0, 0.0)
    return ofs, coef


def resize(src, dsize=None, fx=0.0, fy=0.0, interpolation=INTER_LINEAR):
    """cv2.resize for float32 HxW or HxWxC arrays (dsize = (width, height) like cv2)."""
    src = np.asarray(src)
    if src.dtype != np.float32:
        raise TypeError("oracle resize restates the float32 path only, got %s" % src.dtype)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    sh, sw, cn = src.shape
    if dsize is None or dsize == (0, 0):
        inv_x, inv_y = float(fx), float(fy)
        # saturate_cast<int>(ssize.width * inv_scale_x): round-half-even of a double
        dw, dh = int(np.rint(sw * inv_x)), int(np.rint(sh * inv_y))
    else:
        dw, dh = int(dsize[0]), int(dsize[1])
        inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if interpolation == INTER_LINEAR and scale_x == 2.0 and scale_y == 2.0:
        # resize.cpp (4.2): "if (is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA" -- the exact
        # 2x decimation runs ResizeAreaFast, whose SIMD body ((a+b)+(c+d))*0.25 and scalar tail (((a+b)+c)+d)*0.25
        # associate the four taps differently; which pixels take which depends on the build's SIMD width.  Not restated.
        raise NotImplementedError("cv2.resize INTER_LINEAR at exactly 2x decimation switches to INTER_AREA: not modelled")

    if interpolation == INTER_NEAREST:
        xs = np.minimum(np.floor(np.arange(dw) * (1.0 / inv_x)).astype(np.int64), sw - 1)
        ys = np.minimum(np.floor(np.arange(dh) * (1.0 / inv_y)).astype(np.int64), sh - 1)
        out = src[ys][:, xs]
        return out[:, :, 0] if squeeze else out

    xofs, alpha = axis_tables(sw, dw, scale_x, interpolation)
    yofs, beta = axis_tables(sh, dh, scale_y, interpolation)
    ktaps = alpha.shape[1]
    k2 = ktaps // 2
    if interpolation == INTER_LINEAR:
        xofs, alpha = _linear_x_clamp(xofs, alpha, sw)

    # ---- horizontal pass: rows[sy][dx][c], float32, left-to-right sum of products ----
    # tap j reads column clamp(xofs + j - (k2-1)); for INTER_LINEAR the second tap of a
    # clamped right-border pixel has weight 0 and index sw (clamped to sw-1): product 0.
    cols = np.clip(xofs[:, None] + np.arange(ktaps)[None, :] - (k2 - 1), 0, sw - 1)  # [dw,k]
    hrows = np.zeros((sh, dw, cn), dtype=np.float32)
    for j in range(ktaps):
        term = src[:, cols[:, j], :] * alpha[None, :, j, None]
        hrows = term if j == 0 else (hrows + term)
    hrows = hrows.astype(np.float32)
    if interpolation == INTER_LINEAR:
        # resize.cpp HResizeLinear: for dx >= xmax (sx+1 out of range) D = S[sx]*1
        xmax_mask = (xofs + 1) >= sw
        if xmax_mask.any():
            hrows[:, xmax_mask, :] = src[:, xofs[xmax_mask], :]

    # ---- vertical pass ----
    rows = np.clip(yofs[:, None] + np.arange(ktaps)[None, :] - (k2 - 1), 0, sh - 1)  # [dh,k]
    out = None
    for j in range(ktaps):
        term = hrows[rows[:, j]] * beta[:, j, None, None]
        out = term if j == 0 else (out + term)
    out = out.astype(np.float32)
    return out[:, :, 0] if squeeze else out


def bicubic_x8_at(src2d, py, px, xtab=None):
    """Value of ``resize(src2d, fx=8, fy=8, INTER_CUBIC)[py, px]`` without materialising
    the up-sampled image (what the HIP limb-scoring kernel does).  Scalar float32."""
    sh, sw = src2d.shape
    sx, fxx = _src_coord(px, 0.125)
    sy, fyy = _src_coord(py, 0.125)
    a = cubic_coeffs(fxx)
    b = cubic_coeffs(fyy)
    acc_v = None
    for r in range(4):
        yy = min(max(sy - 1 + r, 0), sh - 1)
        acc_h = None
        for c in range(4):
            xx = min(max(sx - 1 + c, 0), sw - 1)
            t = np.float32(src2d[yy, xx]) * a[c]
            acc_h = t if acc_h is None else np.float32(acc_h + t)
        t = np.float32(acc_h * b[r])
        acc_v = t if acc_v is None else np.float32(acc_v + t)
    return np.float32(acc_v)
