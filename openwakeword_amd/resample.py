"""Rate conversion to 16 kHz for whole batches of streams on the device (SURVEY §8f-3 "on-device resample").

The reference's serving example converts every websocket message with `resampy.resample(data, sample_rate, 16000)` before
`predict()` (examples/web/streaming_server.py:57-58): a stateless, band-limited (Kaiser-windowed sinc) interpolation per
message.  This module designs the equivalent polyphase filter bank once per input rate; `oww_resample` (include/owwhip.h) applies
it to [S, n_in] int16 on the GPU, `apply_numpy` is the host restatement the tests check it against.

    p / q = rate_in / 16000 in lowest terms; output sample j sits at input position j * p / q:
    i0 = (j * p) // q, phase = (j * p) % q,   out[j] = sat_int16(rint(sum_k taps[phase, k] * x[i0 + k - half + 1]))
with x = 0 outside the message.  taps[phase, k] = window((k - half + 1 - phase / q) / (ZC * s)) * sinc((k - half + 1 - phase / q) / s) / s,
s = max(1, rate_in / 16000) (the pass band ends at the lower of the two Nyquist rates), ZC zero crossings each side, every phase
normalised to unit DC gain.
"""
from __future__ import annotations

from fractions import Fraction
from functools import lru_cache
from typing import Tuple

import numpy as np

ZERO_CROSSINGS = 12
KAISER_BETA = 8.0
ROLLOFF = 0.945          # pass band edge as a fraction of the lower Nyquist rate (resampy's kaiser_best uses ~0.9476)


@lru_cache(maxsize=32)
def design(rate_in: int, rate_out: int = 16000) -> Tuple[int, int, np.ndarray]:
    """(p, q, taps float32 [q, 2 * half]) for rate_in -> rate_out."""
    if not 1000 <= int(rate_in) <= 384000:
        raise ValueError(f"sample rate {rate_in} Hz is outside [1000, 384000]")
    r = Fraction(int(rate_in), int(rate_out))
    p, q = r.numerator, r.denominator
    s = max(1.0, rate_in / rate_out) / ROLLOFF
    half = int(np.ceil(ZERO_CROSSINGS * s))
    k = np.arange(2 * half, dtype=np.float64) - (half - 1)                  # tap offsets relative to i0
    ph = np.arange(q, dtype=np.float64)[:, None] / q
    d = k[None, :] - ph                                                     # distance of the tap from the output position, in input samples
    u = d / (ZERO_CROSSINGS * s)
    win = np.where(np.abs(u) < 1.0, np.i0(KAISER_BETA * np.sqrt(np.clip(1.0 - u * u, 0.0, 1.0))) / np.i0(KAISER_BETA), 0.0)
    taps = win * np.sinc(d / s) / s
    taps /= taps.sum(axis=1, keepdims=True)
    return p, q, np.ascontiguousarray(taps, dtype=np.float32)


def n_out_for(n_in: int, rate_in: int, rate_out: int = 16000) -> int:
    p, q, _ = design(rate_in, rate_out)
    return (int(n_in) * q) // p


def apply_numpy(x: np.ndarray, rate_in: int) -> np.ndarray:
    """Host restatement of oww_resample (float64 accumulation): int16 [..., n_in] -> int16 [..., n_out]."""
    p, q, taps = design(rate_in)
    x = np.asarray(x)
    if rate_in == 16000:
        return x.copy()
    n_in = x.shape[-1]
    n_out = (n_in * q) // p
    half = taps.shape[1] // 2
    j = np.arange(n_out)
    i0 = (j * p) // q
    ph = (j * p) % q
    idx = i0[:, None] + np.arange(2 * half)[None, :] - (half - 1)            # [n_out, taps]
    ok = (idx >= 0) & (idx < n_in)
    xs = np.where(ok, x[..., np.clip(idx, 0, n_in - 1)].astype(np.float64), 0.0)
    y = (xs * taps[ph].astype(np.float64)).sum(-1)
    return np.clip(np.rint(y), -32768, 32767).astype(np.int16)
