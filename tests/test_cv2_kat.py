"""Known-answer tests for the OpenCV boundary (oracle/cv2_resize.py), derived WITHOUT numpy float arithmetic.

OpenCV is absent from the image and from the reference tree, so cv2.resize itself cannot be run here.
What can be pinned independently of the numpy restatement:
  * an exact-rational evaluator (`fractions.Fraction`) that follows OpenCV 4.2's scalar code path operation by operation
    (resize.cpp: interpolateCubic with A = -0.75, c3 = 1 - c0 - c1 - c2; source coordinate in double then float;
    HResizeCubic / VResizeCubic: four products summed left to right; HResizeLinear / VResizeLinear) and rounds every
    float32 operation to nearest-even itself -- a hand derivation in software, sharing no code with the oracle;
  * literal answers worked out with it and frozen below (hex floats), so a later edit of either side shows up;
  * analytic properties of the Keys cubic-convolution kernel in exact arithmetic (partition of unity, reproduction of
    constants and of linear ramps) -- they tie the restated coefficients to the published kernel, not to themselves.
The HIP kernels are bit-exact against oracle/cv2_resize.py (tests/test_gpu_parity.py), so these KATs pin them too.
"""
from fractions import Fraction as Fr

import numpy as np
import pytest

from oracle import cv2_resize as cvr


# ---- float32 / float64 rounding of exact rationals (round to nearest, ties to even) ----------------------------------
def _round_to(x, mant_bits, emin):
    if x == 0:
        return Fr(0)
    s = -1 if x < 0 else 1
    a = -x if x < 0 else x
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fr(2) ** e > a:
        e -= 1
    while Fr(2) ** (e + 1) <= a:
        e += 1
    e = max(e, emin)                                   # subnormals share the smallest normal's quantum
    q = Fr(2) ** (e - mant_bits)
    m = a / q
    lo = m.numerator // m.denominator
    rem = m - lo
    if rem > Fr(1, 2) or (rem == Fr(1, 2) and lo % 2 == 1):
        lo += 1
    return s * lo * q


def f32(x):
    return _round_to(Fr(x), 23, -126)


def f64(x):
    return _round_to(Fr(x), 52, -1022)


def as_fr(v):
    return Fr(float(v))                                # exact: every float is a rational


def cubic_coeffs_exact_order(x):
    """interpolateCubic(x, coeffs), float32 operation by operation (imgproc/src/precomp.hpp in 4.2)."""
    A = Fr(-3, 4)
    x1 = f32(x + 1)
    c0 = f32(f32(f32(f32(f32(f32(A * x1) - f32(5 * A)) * x1) + f32(8 * A)) * x1) - f32(4 * A))
    c1 = f32(f32(f32(f32(f32(f32(A + 2) * x) - f32(A + 3)) * x) * x) + 1)
    xm = f32(1 - x)
    c2 = f32(f32(f32(f32(f32(f32(A + 2) * xm) - f32(A + 3)) * xm) * xm) + 1)
    c3 = f32(f32(f32(1 - c0) - c1) - c2)
    return [c0, c1, c2, c3]


def src_coord(d, scale):
    """fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx   (scale is a double, the product too)."""
    f = f32(f64(f64((Fr(d) + Fr(1, 2)) * scale) - Fr(1, 2)))
    s = f.numerator // f.denominator
    return s, f32(f - s)


def resize_cubic_exact_order(src, dw, dh):
    sh, sw = len(src), len(src[0])
    scale_x, scale_y = f64(1 / f64(Fr(dw, sw))), f64(1 / f64(Fr(dh, sh)))
    xs = [src_coord(d, scale_x) for d in range(dw)]
    ys = [src_coord(d, scale_y) for d in range(dh)]
    ax = [cubic_coeffs_exact_order(f) for _, f in xs]
    ay = [cubic_coeffs_exact_order(f) for _, f in ys]
    hrows = []
    for y in range(sh):
        row = []
        for d, (s, _) in enumerate(xs):
            acc = None
            for k in range(4):
                xx = min(max(s - 1 + k, 0), sw - 1)
                t = f32(src[y][xx] * ax[d][k])
                acc = t if acc is None else f32(acc + t)
            row.append(acc)
        hrows.append(row)
    out = []
    for d, (s, _) in enumerate(ys):
        row = []
        for x in range(dw):
            acc = None
            for k in range(4):
                yy = min(max(s - 1 + k, 0), sh - 1)
                t = f32(hrows[yy][x] * ay[d][k])
                acc = t if acc is None else f32(acc + t)
            row.append(acc)
        out.append(row)
    return out


def resize_linear_exact_order(src, dw, dh):
    sh, sw = len(src), len(src[0])
    scale_x, scale_y = f64(1 / f64(Fr(dw, sw))), f64(1 / f64(Fr(dh, sh)))
    out = []
    hrows = []
    for y in range(sh):
        row = []
        for d in range(dw):
            s, f = src_coord(d, scale_x)
            if s < 0:
                s, f = 0, Fr(0)
            if s >= sw - 1:
                s, f = sw - 1, Fr(0)
            if s + 1 >= sw:                                   # HResizeLinear tail: D = S[sx] * 1
                row.append(f32(src[y][s] * 1))
            else:
                row.append(f32(f32(src[y][s] * f32(1 - f)) + f32(src[y][s + 1] * f)))
        hrows.append(row)
    for d in range(dh):
        s, f = src_coord(d, scale_y)
        y0, y1 = min(max(s, 0), sh - 1), min(max(s + 1, 0), sh - 1)
        b0, b1 = f32(1 - f), f
        out.append([f32(f32(hrows[y0][x] * b0) + f32(hrows[y1][x] * b1)) for x in range(dw)])
    return out


def _hex(fr):
    return float(fr).hex()


# ---- the tests ------------------------------------------------------------------------------------------------------
def test_rounding_helper_matches_ieee():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.standard_normal(2)
        x, y = np.float32(a), np.float32(b)
        assert f32(as_fr(x) * as_fr(y)) == as_fr(x * y)
        assert f32(as_fr(x) + as_fr(y)) == as_fr(x + y)
    assert f32(Fr(1, 3)) == as_fr(np.float32(1.0) / np.float32(3.0))
    assert f32(Fr(2) ** -140 * 3) == as_fr(np.float32(3 * 2.0 ** -140))       # subnormal


def test_cubic_coefficients_the_eight_x8_phases_known_answers():
    """x8 up-sampling only ever uses the fractional offsets (2p+1)/16.  Frozen answers (float32, hex) from the
    operation-by-operation derivation; the oracle and the HIP kernels' table (pn_debug_cubic_coeffs) must hit them."""
    frozen = {
        1: ['-0x1.5180000000000p-5', '0x1.fba8000000000p-1', '0x1.ad80000000000p-5', '-0x1.6800000000000p-9'],      # x = 1/16: c0 = -0.041199, c1 = 0.991516 (worked by hand in the docstring's order)
        7: ['-0x1.a940000000000p-4', '0x1.5918000000000p-1', '0x1.0568000000000p-1', '-0x1.4ac0000000000p-4'],      # x = 7/16
    }
    for p in range(8):
        x = Fr(2 * p + 1, 16)
        want = cubic_coeffs_exact_order(x)
        got = cvr.cubic_coeffs(np.float32(float(x)))
        assert [as_fr(v) for v in got] == want, p
        if 2 * p + 1 in frozen:
            assert [_hex(v) for v in want] == frozen[2 * p + 1], (p, [_hex(v) for v in want])
    # and against exact (unrounded) Keys kernel values: the float32 evaluation order costs at most a few ulp
    for p in range(8):
        x = Fr(2 * p + 1, 16)
        A = Fr(-3, 4)
        exact = [((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A, ((A + 2) * x - (A + 3)) * x * x + 1,
                 ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1]
        exact.append(1 - sum(exact))
        assert sum(exact) == 1
        for e, g in zip(exact, cubic_coeffs_exact_order(x)):
            assert abs(e - g) <= Fr(1, 2 ** 22)


def test_keys_kernel_reproduces_constants_and_ramps_in_exact_arithmetic():
    A = Fr(-3, 4)
    for x in (Fr(1, 16), Fr(5, 16), Fr(1, 2), Fr(15, 16)):
        c = [((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A, ((A + 2) * x - (A + 3)) * x * x + 1,
             ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1]
        c.append(1 - sum(c))
        assert sum(c) == 1                                                   # constants are reproduced (partition of unity)
        # mirror symmetry of the kernel: the taps at offset 1 - x are the taps at x reversed
        xr = 1 - x
        cr = [((A * (xr + 1) - 5 * A) * (xr + 1) + 8 * A) * (xr + 1) - 4 * A, ((A + 2) * xr - (A + 3)) * xr * xr + 1,
              ((A + 2) * (1 - xr) - (A + 3)) * (1 - xr) * (1 - xr) + 1]
        cr.append(1 - sum(cr))
        assert cr == c[::-1]
        # A = -0.75 (OpenCV) is NOT Keys' third-order choice (-0.5): a linear ramp is reproduced only up to 2 (A + 1/2) x (1 - x) (2x - 1)
        assert sum(ck * (k - 1) for k, ck in enumerate(c)) - x == 2 * (A + Fr(1, 2)) * x * (1 - x) * (2 * x - 1)


def test_bicubic_3x3_to_24x24_matches_the_derivation_bit_for_bit():
    src = [[Fr(1, 8), Fr(3, 4), Fr(-1, 2)], [Fr(5, 16), Fr(-7, 8), Fr(9, 16)], [Fr(1), Fr(1, 4), Fr(-3, 8)]]
    want = resize_cubic_exact_order(src, 24, 24)
    arr = np.array([[float(v) for v in r] for r in src], dtype=np.float32)
    got = cvr.resize(arr, fx=8, fy=8, interpolation=cvr.INTER_CUBIC)
    assert got.shape == (24, 24)
    for y in range(24):
        for x in range(24):
            assert as_fr(got[y, x]) == want[y][x], (y, x)
            assert as_fr(cvr.bicubic_x8_at(arr, y, x)) == want[y][x], (y, x)
    # frozen corner / centre / edge answers of this patch (hex float32)
    frozen = {(0, 0): '0x1.59d9f80000000p-6', (11, 12): '-0x1.a297b80000000p-1', (23, 23): '-0x1.1e70620000000p-1', (4, 19): '-0x1.8d4b7a0000000p-2'}
    for (y, x), h in frozen.items():
        assert _hex(want[y][x]) == h, ((y, x), _hex(want[y][x]))


def test_bilinear_downscale_480_to_224_columns_and_upscale_borders():
    rng = np.random.default_rng(5)
    # down-scaling rows of the real geometry (480 -> 224 columns, 640 -> 224 rows on a thin slab keeps the test fast)
    src = rng.uniform(0, 6, (9, 480)).astype(np.float16).astype(np.float32)
    got = cvr.resize(src, (224, 3), interpolation=cvr.INTER_LINEAR)
    want = resize_linear_exact_order([[as_fr(v) for v in r] for r in src], 224, 3)
    for y in range(3):
        for x in list(range(0, 224, 17)) + [0, 1, 222, 223]:
            assert as_fr(got[y, x]) == want[y][x], (y, x)
    # up-scaling: the first / last destination columns clamp to the border pixel with weight 1 (resize.cpp: sx < 0 / sx >= W-1)
    small = rng.uniform(0, 6, (5, 7)).astype(np.float32)
    got = cvr.resize(small, (29, 11), interpolation=cvr.INTER_LINEAR)
    want = resize_linear_exact_order([[as_fr(v) for v in r] for r in small], 29, 11)
    for y in range(11):
        for x in range(29):
            assert as_fr(got[y, x]) == want[y][x], (y, x)
    assert np.array_equal(got[0, :2], small[0, [0, 0]])
    assert got[0, 28] == small[0, 6]


def test_exact_2x_decimation_is_refused_loudly():
    """cv::resize switches INTER_LINEAR to INTER_AREA when both scale factors are exactly 2 (resize.cpp 4.2:
    `if (is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA`), whose SIMD body and scalar tail
    associate the four taps differently.  The restatement does not model that branch: it must refuse, not guess."""
    with pytest.raises(NotImplementedError):
        cvr.resize(np.zeros((448, 448), np.float32), (224, 224), interpolation=cvr.INTER_LINEAR)
    # one exact factor of 2 only (480 x 448 -> 224 x 224) stays on the linear path
    cvr.resize(np.zeros((448, 480), np.float32), (224, 224), interpolation=cvr.INTER_LINEAR)
