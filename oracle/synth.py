"""Deterministic synthetic activations / weights for parity tests, goldens and the bench.

TEST INFRASTRUCTURE (part of ``oracle/``): only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package.  The product path
(``whisper-sae_amd/``) never does.

Everything here is integer arithmetic plus IEEE-754 correctly rounded float64 ops, so the streams
are bit-identical on every machine and every numpy/torch version (numpy's ``Generator``
distributions carry no cross-version guarantee, torch's RNG neither).  The values are the
stand-in for Whisper encoder activations that SURVEY.md row D asks for (``x ~ N(0,1)`` i.i.d.,
seed 42 as ``configs/*.yaml: training.seed``): an Irwin-Hall sum of four 16-bit uniforms is used
instead of a Box-Muller transform so that no libm transcendental enters the stream.
"""

from __future__ import annotations

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        return z ^ (z >> np.uint64(31))


def counter_u64(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """``n`` 64-bit words: word ``i`` depends only on ``(seed, stream, i)``."""
    key = _splitmix64(np.array([seed], dtype=np.uint64))[0]
    key = _splitmix64(np.array([key ^ np.uint64(stream * 0x100000001B3 + 1)], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    return _splitmix64(ctr & _MASK64)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bfloat16 (ties to even); result returned as float32.

    Integer restatement of the hardware conversion (``v_cvt_pk_bf16_f32``) for finite inputs.
    """
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return (r & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32).reshape(a.shape)


def bf16_bits(a: np.ndarray) -> np.ndarray:
    """uint16 bit pattern of ``bf16_round(a)`` (what lives in the on-device ring buffer)."""
    return (bf16_round(a).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def normal(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Approximately N(0,1) float32 values (Irwin-Hall, 4 x u16), deterministic everywhere."""
    n = int(np.prod(shape))
    w = counter_u64(n, seed, stream)
    s = ((w & np.uint64(0xFFFF)) + ((w >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((w >> np.uint64(32)) & np.uint64(0xFFFF)) + (w >> np.uint64(48))).astype(np.float64)
    # sum of four U{0..65535}: mean 131070, variance 4*(65536^2-1)/12
    sd = np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)
    return ((s - 131070.0) / sd).astype(np.float32).reshape(shape)


def uniform(shape, seed: int, stream: int = 0, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """Uniform float32 in [lo, hi) from the top 24 bits of each counter word."""
    n = int(np.prod(shape))
    w = counter_u64(n, seed, stream)
    u = (w >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def activations(n_rows: int, dim: int, seed: int = 42, stream: int = 0, bf16: bool = True) -> np.ndarray:
    """Synthetic activation rows ``[n_rows, dim]`` (bf16-representable when ``bf16``)."""
    x = normal((n_rows, dim), seed, stream)
    return bf16_round(x) if bf16 else x


def sae_weights(input_dim: int, hidden_dim: int, seed: int = 42, bf16: bool = True,
                b_pre_scale: float = 0.0, bias_scale: float | None = None) -> dict:
    """A TopKSAE parameter set shaped like the reference's initialisation (model.py:63-89).

    encoder.weight / encoder.bias ~ U(-1/sqrt(D), 1/sqrt(D)) (``nn.Linear`` default);
    decoder.weight: random direction per column, unit L2 norm, times 0.1 (``_init_decoder``);
    decoder.bias ~ U(-1/sqrt(H), 1/sqrt(H)); ``b_pre`` zeros unless ``b_pre_scale`` is given.
    Values come from the counter generator (not torch's RNG) so fixtures are version-proof; with
    ``bf16`` every tensor is rounded to bf16-representable float32.
    """
    d, h = input_dim, hidden_dim
    be = 1.0 / np.sqrt(d) if bias_scale is None else bias_scale
    w = {
        "encoder.weight": uniform((h, d), seed, 11, -1.0 / np.sqrt(d), 1.0 / np.sqrt(d)),
        "encoder.bias": uniform((h,), seed, 12, -be, be),
        "decoder.bias": uniform((d,), seed, 14, -1.0 / np.sqrt(h), 1.0 / np.sqrt(h)),
        "b_pre": (normal((d,), seed, 15) * np.float32(b_pre_scale)).astype(np.float32),
    }
    wd = normal((d, h), seed, 13).astype(np.float64)
    wd = wd / np.maximum(np.sqrt((wd * wd).sum(axis=0, keepdims=True)), 1e-12) * 0.1
    w["decoder.weight"] = wd.astype(np.float32)
    if bf16:
        w = {k: bf16_round(v) for k, v in w.items()}
    return w


def topk_margin(pre: np.ndarray, k: int) -> np.ndarray:
    """Per-row relative gap between the k-th and (k+1)-th largest pre-activation.

    Fixtures assert this is comfortably above fp32 summation-order noise: only
    then is "TopK index sets bit-exact" a meaningful statement about two different summation orders.
    """
    s = -np.sort(-pre.astype(np.float64), axis=1)
    if k >= pre.shape[1]:
        return np.full(pre.shape[0], np.inf)
    return (s[:, k - 1] - s[:, k]) / np.maximum(np.abs(s[:, k - 1]), 1e-30)
