"""TEST INFRASTRUCTURE ONLY -- CPU restatement of `audio::resample` (mlx-rs-core/src/audio.rs:178-277).

PARITY UNPINNED for sample values: the arithmetic lives in the third-party crate `rubato` (mlx-rs-core/Cargo.toml:21 `rubato = "0.14"`,
no Cargo.lock in the tree, not vendored, not buildable here -- no Rust toolchain).  What is restated below is rubato 0.14's published
algorithm for exactly the configuration the reference requests (:186-203):
    SincFixedIn::<f32>::new(ratio, 2.0, {sinc_len 256, f_cutoff 0.95, Cubic, oversampling 256, BlackmanHarris2}, chunk, 1)
      * sinc table (rubato `make_sincs`): 256 x 256 taps of window(x) * sinc((x - N/2) * cutoff / 256), N = 65536, computed in f32,
        normalised by sum / 256, stored phase-reversed (`sincs[factor - n - 1][p] = y[factor * p + n]`); cutoff = 0.95 when
        upsampling and 0.95 * ratio when downsampling; window = squared 4-term Blackman-Harris (periodic form, x / N)
      * `process_into_buffer`: a buffer of chunk + 2 * 256 samples whose last 512 samples carry over; a float64 read position `idx`
        starting at -128 that advances by 1 / ratio per output while idx < chunk - 257 - ceil(1 / ratio); for each output the 4
        nearest (sample, phase) points around idx, a 256-tap dot product each, cubic interpolation at the fractional phase
      * `process_partial(Some(x))` = the chunk zero-padded; `process_partial(Some(&[]))` = an all-zero chunk
and the reference's own driver around it (:205-277), which IS in the tree and is restated line by line: full chunks of min(4096, len),
the zero-padded remainder of which only ceil(remaining * ratio) outputs are kept (:239-256), one all-zero flush chunk (:258-267),
truncation to round(len * ratio) (:269-273).  Anchors available without rubato: the reference's tests (length only: audio.rs:705-710,
funasr-qwen4b-mlx/src/audio.rs:692-706), the identity for equal rates / empty input (:179-181), and the mathematical property a
windowed-sinc resampler must have (a band-limited sine comes out as the same sine at the new rate) -- tests/test_resample.py.
"""
from __future__ import annotations

import math

import numpy as np

SINC_LEN, OVERSAMPLING, F_CUTOFF, MAX_CHUNK = 256, 256, 0.95, 4096
f32 = np.float32


def blackman_harris2(npoints: int) -> np.ndarray:
    """rubato windows.rs: squared 4-term Blackman-Harris, periodic (x / npoints), evaluated in f32."""
    x = np.arange(npoints, dtype=f32)
    n = f32(npoints)
    pi2, pi4, pi6 = f32(2 * math.pi), f32(4 * math.pi), f32(6 * math.pi)
    w = (f32(0.35875) - f32(0.48829) * np.cos(pi2 * x / n, dtype=f32) + f32(0.14128) * np.cos(pi4 * x / n, dtype=f32)
         - f32(0.01168) * np.cos(pi6 * x / n, dtype=f32)).astype(f32)
    return (w * w).astype(f32)


def make_sincs(npoints: int, factor: int, f_cutoff: float) -> np.ndarray:
    """rubato sinc.rs `make_sincs`: [factor, npoints] f32; row j holds the kernel sampled at p - npoints/2 + (factor - 1 - j) / factor."""
    tot = npoints * factor
    x = np.arange(tot, dtype=f32)
    arg = ((x - f32(tot // 2)) * f32(f_cutoff) / f32(factor)).astype(f32)
    a = (arg * f32(math.pi)).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(arg == 0, f32(1.0), np.sin(a, dtype=f32) / a).astype(f32)
    y = (blackman_harris2(tot) * s).astype(f32)
    total = np.cumsum(y, dtype=f32)[-1] / f32(factor)          # `sum += val` in f32, sequentially
    y = (y / total).astype(f32)
    return np.ascontiguousarray(y.reshape(npoints, factor).T[::-1])   # sincs[factor - n - 1][p] = y[factor * p + n]


def plan(n_in: int, src_rate: int, dst_rate: int):
    """The read positions the reference's driver makes rubato visit: per kept output (stream index of the first of the 4 points,
    its phase, the cubic fraction).  Stream = the samples followed by zeros (padded remainder + flush chunk); position g = chunk * c +
    idx is continuous over chunks, but idx itself is re-based every chunk exactly as rubato does (`last_index = idx - chunk`)."""
    ratio = float(dst_rate) / float(src_rate)
    chunk = min(MAX_CHUNK, n_in)
    t_ratio = 1.0 / ratio
    end_idx = chunk - (SINC_LEN + 1) - math.ceil(t_ratio)
    n_full, remaining = divmod(n_in, chunk)
    keep = [None] * n_full                                     # None = every output of the chunk
    if remaining:
        keep.append(math.ceil(remaining * ratio))              # :250-252
    keep.append(None)                                          # the flush chunk, :258-267
    g0, sub0, frac = [], [], []
    idx = -float(SINC_LEN // 2)
    for c, lim in enumerate(keep):
        n = 0
        while idx < end_idx:
            idx += t_ratio
            if lim is None or n < lim:
                fl = math.floor(idx)
                index, sub = int(fl), int(math.floor((idx - fl) * OVERSAMPLING))
                sub -= 1                                       # get_nearest_times_4: points[0] is one phase step before (index, sub)
                if sub < 0:
                    sub += OVERSAMPLING
                    index -= 1
                g0.append(c * chunk + index)
                sub0.append(sub)
                v = idx * OVERSAMPLING
                frac.append(v - math.floor(v))
            n += 1
        idx -= chunk
    expected = int(round(n_in * ratio))                        # :270 (f64 round: half away from zero)
    expected = int(math.floor(n_in * ratio + 0.5)) if n_in * ratio >= 0 else expected
    if len(g0) > expected:
        g0, sub0, frac = g0[:expected], sub0[:expected], frac[:expected]
    return np.array(g0, np.int64), np.array(sub0, np.int64), np.array(frac, np.float64), ratio


def resample(samples, src_rate: int, target_rate: int) -> np.ndarray:
    """audio::resample (:178-277).  float32 in, float32 out."""
    x = np.asarray(samples, dtype=f32).ravel()
    if src_rate == target_rate or x.size == 0:                 # :179-181
        return x.copy()
    g0, sub0, frac, ratio = plan(x.size, src_rate, target_rate)
    cutoff = F_CUTOFF if ratio >= 1.0 else f32(F_CUTOFF) * f32(ratio)
    sincs = make_sincs(SINC_LEN, OVERSAMPLING, float(cutoff))
    lo, hi = int(g0.min()) - 1, int(g0.max()) + SINC_LEN + 8
    stream = np.zeros(hi - lo, f32)                            # samples followed by zeros; zeros before the start as well
    a, b = max(lo, 0), min(hi, x.size)
    if b > a:
        stream[a - lo:b - lo] = x[a:b]
    pts = np.empty((4, g0.size), np.float64)
    taps = np.arange(SINC_LEN)
    for k in range(4):                                         # the 4 consecutive (sample, phase) points
        sub = sub0 + k
        idx = g0 + sub // OVERSAMPLING
        sub = sub % OVERSAMPLING
        for s0 in range(0, g0.size, 1 << 14):                  # blocks: [block, 256] gathers
            sl = slice(s0, s0 + (1 << 14))
            seg = stream[(idx[sl, None] - lo) + taps[None, :]]
            pts[k, sl] = np.einsum("ij,ij->i", seg.astype(np.float64), sincs[sub[sl]].astype(np.float64))
    y0, y1, y2, y3 = (pts[k].astype(f32) for k in range(4))
    xf = frac.astype(f32)
    # rubato `interp_cubic`: y1 at x = 0, y2 at x = 1
    a0 = y1
    a1 = -(f32(1.0) / f32(3.0)) * y0 - f32(0.5) * y1 + y2 - (f32(1.0) / f32(6.0)) * y3
    a2 = f32(0.5) * (y0 + y2) - y1
    a3 = f32(0.5) * (y1 - y2) + (f32(1.0) / f32(6.0)) * (y3 - y0)
    x2 = xf * xf
    return (a0 + a1 * xf + a2 * x2 + a3 * x2 * xf).astype(f32)


def output_time(n, ratio: float) -> np.ndarray:
    """Input-sample time the n-th output of a long signal represents (derived from the table layout: a (sample i, phase j) point
    is centred on stream position i + 127 + (j + 1) / 256, and idx starts at -128): (n + 1) / ratio - 1 + 1 / 256."""
    return (np.asarray(n, np.float64) + 1.0) / ratio - 1.0 + 1.0 / OVERSAMPLING
