"""CPU oracle: numpy restatement of the reference's SAE train step.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; it is the checker, never the thing measured as the product and
never a fallback for it (the product path in ``whisper-sae_amd/`` raises when the HIP library is
missing).

Pinned (tests/test_oracle_golden.py) against golden vectors produced *in the build container* by
importing the real reference (``tests/golden/make_golden.py``); the reference ships no golden
vectors of its own, its property tests are mirrored in tests/ as well.

Every function cites the reference lines it restates (paths relative to the reference repo).
Arithmetic lives in PyTorch in the reference (un-vendored, ``torch>=2.1`` unpinned); the torch op
semantics restated here are: ``nn.Linear`` (x @ W.T + b), ``torch.topk`` (k largest, sorted
descending), ``scatter_``, ``F.mse_loss`` (mean over all elements), ``F.normalize`` (x / max(||x||,
eps), eps=1e-12), ``clip_grad_norm_`` (coef = max_norm / (total_norm + 1e-6), clamped to 1),
``AdamW`` (decoupled decay, bias-corrected, eps outside the sqrt), ``LinearLR`` /
``CosineAnnealingLR`` / ``SequentialLR``.

Two arithmetic modes, matching the two modes of the HIP path:

* ``"fp32"`` -- the reference CPU semantics: everything in float32 (intermediates here are
  float64 and rounded at the points where the reference stores a float32 tensor).
* ``"amp"`` -- what the MI355X path computes under ``use_amp`` (bf16 MFMA operands, fp32
  accumulate): ``pre = bf16(W_e) @ bf16(x) + (b_e - bf16(W_e) @ b_pre)``; decode and ``dh`` gather rows
  of the bf16 shadow of ``W_d`` and accumulate in fp32 (TopK values, residual, loss, ``g`` stay
  fp32); weight gradients from bf16-rounded ``hidden``, ``g``, ``dpre``, ``x``.  On inputs
  whose ``x``/``W_e`` are bf16-representable the two modes agree in the forward pass to fp32
  rounding, which is how "identical inputs" parity with the reference is defined.

TopK tie rule (the reference leaves it to ``torch.topk``, whose CPU tie order is implementation
defined): value descending, then **lowest index first**.  Fixtures are generated tie-free with an
asserted k/k+1 margin.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .synth import bf16_round

F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------------------------
# state
# --------------------------------------------------------------------------------------------
@dataclass
class SAEState:
    """Parameters + buffers of a TopKSAE (state-dict keys of model.py:63-77) and AdamW moments."""

    W_e: np.ndarray  # encoder.weight [H, D]
    b_e: np.ndarray  # encoder.bias   [H]
    W_d: np.ndarray  # decoder.weight [D, H]
    b_d: np.ndarray  # decoder.bias   [D]
    b_pre: np.ndarray  # [D]
    k: int
    dead_feature_threshold: int = 10_000
    last_activated: np.ndarray = None  # feature_last_activated int64 [H]
    step_count: int = 0
    adam_m: dict = field(default_factory=dict)
    adam_v: dict = field(default_factory=dict)
    adam_t: int = 0

    PARAMS = ("W_e", "b_e", "W_d", "b_d", "b_pre")

    def __post_init__(self):
        for n in self.PARAMS:
            setattr(self, n, np.array(getattr(self, n), dtype=F32, copy=True))
        if self.last_activated is None:
            self.last_activated = np.zeros(self.W_e.shape[0], dtype=np.int64)
        else:
            self.last_activated = np.array(self.last_activated, dtype=np.int64, copy=True)

    @classmethod
    def from_state_dict(cls, sd: dict, k: int, dead_feature_threshold: int = 10_000) -> "SAEState":
        g = lambda n: np.asarray(sd[n])  # noqa: E731
        st = cls(g("encoder.weight"), g("encoder.bias"), g("decoder.weight"), g("decoder.bias"),
                 g("b_pre"), k, dead_feature_threshold)
        if "feature_last_activated" in sd:
            st.last_activated = np.array(sd["feature_last_activated"], dtype=np.int64)
        if "step_count" in sd:
            st.step_count = int(np.asarray(sd["step_count"]))
        return st

    def state_dict(self) -> dict:
        return {"encoder.weight": self.W_e, "encoder.bias": self.b_e, "decoder.weight": self.W_d,
                "decoder.bias": self.b_d, "b_pre": self.b_pre,
                "feature_last_activated": self.last_activated,
                "step_count": np.int64(self.step_count)}

    def copy(self) -> "SAEState":
        c = SAEState(self.W_e, self.b_e, self.W_d, self.b_d, self.b_pre, self.k,
                     self.dead_feature_threshold, self.last_activated, self.step_count)
        c.adam_m = {n: v.copy() for n, v in self.adam_m.items()}
        c.adam_v = {n: v.copy() for n, v in self.adam_v.items()}
        c.adam_t = self.adam_t
        return c


# --------------------------------------------------------------------------------------------
# forward  (model.py:98-166)
# --------------------------------------------------------------------------------------------
def pre_activation(st: SAEState, x: np.ndarray, mode: str = "fp32") -> np.ndarray:
    """model.py:108-111: ``encoder(x - b_pre)``."""
    x = np.asarray(x, dtype=F32)
    if mode == "fp32":
        xc = (x - st.b_pre).astype(F32)
        return (xc.astype(F64) @ st.W_e.astype(F64).T + st.b_e.astype(F64)).astype(F32)
    if mode == "amp":
        w = bf16_round(st.W_e).astype(F64)
        c = (st.b_e.astype(F64) - w @ st.b_pre.astype(F64)).astype(F32)  # folded bias, fp32 on device
        return (bf16_round(x).astype(F64) @ w.T + c.astype(F64)).astype(F32)
    raise ValueError(mode)


def topk_select(pre: np.ndarray, k: int):
    """model.py:114 ``torch.topk(pre, k, dim=-1)``: values sorted descending + indices.

    Tie rule of this build: equal values are ordered by ascending index.
    """
    order = np.argsort(-pre.astype(F64), axis=1, kind="stable")[:, :k]
    vals = np.take_along_axis(pre, order, axis=1)
    return vals.astype(F32), order.astype(np.int64)


def check_selection(pre: np.ndarray, idx: np.ndarray, k: int, rtol: float = 1e-5) -> np.ndarray:
    """Per-row verdict: is ``idx`` a valid TopK-k index set of ``pre`` up to a relative slack ``rtol``?

    Valid means: k distinct indices; every element clearly above the k-th largest value is selected;
    nothing clearly below it is.  "Clearly" = by more than ``rtol * |k-th value|`` (fp32 accumulation
    of a length-D dot product in a different order moves a value by ~1e-7 relative).
    """
    pre64 = pre.astype(F64)
    B, H = pre64.shape
    idx = np.asarray(idx, dtype=np.int64)
    kth = -np.sort(-pre64, axis=1)[:, k - 1]
    slack = rtol * np.maximum(np.abs(kth), 1e-30)
    sel = np.zeros((B, H), dtype=bool)
    np.put_along_axis(sel, idx, True, axis=1)
    distinct = sel.sum(axis=1) == k
    must = pre64 > (kth + slack)[:, None]
    may = pre64 >= (kth - slack)[:, None]
    return distinct & ~(must & ~sel).any(axis=1) & ~(sel & ~may).any(axis=1)


def densify(vals: np.ndarray, idx: np.ndarray, hidden_dim: int) -> np.ndarray:
    """model.py:115-116: ``zeros_like(pre).scatter_(-1, idx, relu(vals))``."""
    hidden = np.zeros((vals.shape[0], hidden_dim), dtype=F32)
    np.put_along_axis(hidden, idx, np.maximum(vals, 0).astype(F32), axis=1)
    return hidden


def decode(st: SAEState, hidden: np.ndarray, mode: str = "fp32") -> np.ndarray:
    """model.py:129: ``decoder(hidden) + b_pre`` (``"amp"``: through the bf16 shadow of W_d)."""
    w_d = bf16_round(st.W_d) if mode == "amp" else st.W_d
    return (hidden.astype(F64) @ w_d.astype(F64).T + st.b_d.astype(F64)
            + st.b_pre.astype(F64)).astype(F32)


def forward(st: SAEState, x: np.ndarray, mode: str = "fp32", training: bool = True,
            select: np.ndarray | None = None) -> dict:
    """model.py:131-166 ``TopKSAE.forward``; updates dead tracking when ``training``.

    ``select`` (tests only): an index array ``[B, k]`` to use in place of ``topk_select``.  Large-batch
    parity tests pass the device's selection for the few rows whose k-th / (k+1)-th pre-activations
    are closer than fp32 summation-order noise (after checking with ``check_selection`` that it IS a
    valid TopK of those rows up to that noise), so that everything downstream compares element-wise.
    """
    x = np.asarray(x, dtype=F32)
    pre = pre_activation(st, x, mode)
    if select is None:
        vals, idx = topk_select(pre, st.k)
    else:
        idx = np.asarray(select, dtype=np.int64)
        vals = np.take_along_axis(pre, idx, axis=1).astype(F32)
    hidden = densify(vals, idx, st.W_e.shape[0])
    recon = decode(st, hidden, mode)
    resid = recon.astype(F64) - x.astype(F64)
    loss = F32(np.mean(resid * resid))  # F.mse_loss, mean over B*D (model.py:145)
    l0 = F32((hidden > 0).sum(axis=1).astype(F64).mean())  # model.py:148
    if training:
        update_dead_features(st, hidden)
    return {"pre": pre, "vals": vals, "idx": idx, "hidden": hidden, "reconstructed": recon,
            "loss": loss, "reconstruction_loss": loss, "sparsity_loss": F32(0.0), "l0": l0}


def update_dead_features(st: SAEState, hidden: np.ndarray) -> None:
    """model.py:168-181: ``step_count += 1; last_activated[(hidden>0).any(0)] = step_count``."""
    st.step_count += 1
    fired = (hidden > 0).any(axis=0)
    st.last_activated[fired] = st.step_count


def dead_mask(st: SAEState) -> np.ndarray:
    """model.py:183-190: strict ``>`` on ``step_count - last_activated``."""
    return (st.step_count - st.last_activated) > st.dead_feature_threshold


def dead_ratio(st: SAEState) -> float:
    """model.py:192-195."""
    return float(dead_mask(st).astype(F32).mean())


# --------------------------------------------------------------------------------------------
# backward  (autograd of model.py:131-154 through training.py:184)
# --------------------------------------------------------------------------------------------
def backward(st: SAEState, x: np.ndarray, fwd: dict, mode: str = "fp32") -> dict:
    """Gradients of ``loss = mean((recon - x)^2)`` w.r.t. the five parameter tensors.

    g = 2 (recon - x) / (B D);  dW_d = g^T hidden;  db_d = sum_b g;  dh = g W_d;
    dpre = dh * 1[selected and v > 0];  dW_e = dpre^T (x - b_pre);  db_e = sum_b dpre;
    db_pre = sum_b g - sum_b dpre W_e.
    In ``"amp"`` mode the two weight-gradient contractions use bf16-rounded operands (what the MFMA
    kernels are fed), ``dpre`` is rounded to bf16 once where it is produced, and the ``b_pre`` path
    goes through the bf16 encoder weights, mirroring the folded bias of ``pre_activation``.
    """
    x = np.asarray(x, dtype=F32)
    B, D = x.shape
    hidden = fwd["hidden"].astype(F64)
    g = (2.0 * (fwd["reconstructed"].astype(F64) - x.astype(F64)) / (B * D)).astype(F32)
    g64 = g.astype(F64)
    db_d = g64.sum(axis=0)
    if mode == "amp":  # bf16 decoder shadow against bf16(g): what the dot2 instruction is fed
        dh = bf16_round(g).astype(F64) @ bf16_round(st.W_d).astype(F64)
    else:
        dh = g64 @ st.W_d.astype(F64)  # [B, H]
    dpre = np.where(hidden > 0, dh, 0.0)
    if mode == "fp32":
        xc = (x - st.b_pre).astype(F64)
        dW_d = g64.T @ hidden
        dW_e = dpre.T @ xc
        db_e = dpre.sum(axis=0)
        db_pre = db_d - (dpre @ st.W_e.astype(F64)).sum(axis=0)
    elif mode == "amp":
        dpre = bf16_round(dpre.astype(F32)).astype(F64)
        w = bf16_round(st.W_e).astype(F64)
        dW_d = bf16_round(g).astype(F64).T @ bf16_round(hidden.astype(F32)).astype(F64)
        db_e = dpre.sum(axis=0)
        dW_e = dpre.T @ bf16_round(x).astype(F64) - np.outer(db_e, st.b_pre.astype(F64))
        db_pre = db_d - db_e @ w
    else:
        raise ValueError(mode)
    return {"W_e": dW_e.astype(F32), "b_e": db_e.astype(F32), "W_d": dW_d.astype(F32),
            "b_d": db_d.astype(F32), "b_pre": db_pre.astype(F32), "g": g,
            "dpre": dpre.astype(F32)}


# --------------------------------------------------------------------------------------------
# optimizer tail  (training.py:186-202)
# --------------------------------------------------------------------------------------------
def grad_total_norm(grads: dict) -> float:
    """``clip_grad_norm_``: L2 norm over all gradient tensors together."""
    return math.sqrt(sum(float((grads[n].astype(F64) ** 2).sum()) for n in SAEState.PARAMS))


def clip_coef(total_norm: float, max_norm: float) -> float:
    """training.py:188-191: ``min(1, max_norm / (total_norm + 1e-6))``."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def adamw_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.AdamW single-tensor update (training.py:63-67, :193).

    p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2;
    denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (lr / (1-b1^t)) * m / denom.
    """
    p64, g64 = p.astype(F64), g.astype(F64)
    p64 = p64 * (1.0 - lr * weight_decay)
    m64 = m.astype(F64) + (g64 - m.astype(F64)) * (1.0 - beta1)
    v64 = beta2 * v.astype(F64) + (1.0 - beta2) * g64 * g64
    m32, v32 = m64.astype(F32), v64.astype(F32)
    bc1 = 1.0 - beta1 ** t
    bc2_sqrt = math.sqrt(1.0 - beta2 ** t)
    denom = np.sqrt(v32.astype(F64)) / bc2_sqrt + eps
    p64 = p64 - (lr / bc1) * (m32.astype(F64) / denom)
    return p64.astype(F32), m32, v32


def normalize_decoder(W_d: np.ndarray) -> np.ndarray:
    """model.py:91-96: ``F.normalize(W_d, dim=0)`` -- each column / max(||col||_2, 1e-12)."""
    n = np.sqrt((W_d.astype(F64) ** 2).sum(axis=0, keepdims=True))
    return (W_d.astype(F64) / np.maximum(n, 1e-12)).astype(F32)


def lr_sequence(n: int, base_lr: float, warmup_steps_cfg: int, total_steps: int) -> list:
    """Learning rates in effect for optimizer steps 0..n-1 (training.py:136-159).

    ``warmup = min(cfg.warmup_steps, total//10)``; ``LinearLR(0.01 -> 1.0, warmup)`` then
    ``CosineAnnealingLR(T_max = total - warmup, eta_min = 0.1 lr)`` joined by
    ``SequentialLR(milestones=[warmup])``.  This restates torch's *chained (recursive)* update
    rules rather than the closed forms, because the reference inherits their corner cases: with
    ``warmup == 0`` (``total < 10`` or ``warmup_steps: 0``) the linear phase never runs, the rate
    starts at ``0.01 lr`` and the cosine recursion is applied to that value (golden set G5
    "nowarm"/"tiny").
    """
    warm = min(warmup_steps_cfg, total_steps // 10)
    t_max = total_steps - warm
    eta_min = 0.1 * base_lr
    lr = base_lr * 0.01  # LinearLR initial step (start_factor)
    out = [lr]
    lin_e, cos_e = 0, -1
    for seq_e in range(1, n):
        if seq_e < warm:  # bisect_right([warm], seq_e) == 0 -> LinearLR.step()
            lin_e += 1
            lr = lr * (1.0 + 0.99 / (warm * 0.01 + (lin_e - 1) * 0.99))
        elif seq_e == warm:  # milestone: CosineAnnealingLR closed form at epoch 0
            cos_e = 0
            lr = eta_min + (base_lr - eta_min) * (1.0 + math.cos(0.0)) / 2.0
        else:  # CosineAnnealingLR.step(), chained form
            cos_e += 1
            if (cos_e - 1 - t_max) % (2 * t_max) == 0:
                lr = lr + (base_lr - eta_min) * (1.0 - math.cos(math.pi / t_max)) / 2.0
            else:
                lr = ((1.0 + math.cos(math.pi * cos_e / t_max))
                      / (1.0 + math.cos(math.pi * (cos_e - 1) / t_max)) * (lr - eta_min) + eta_min)
        out.append(lr)
    return out


def lr_at(step: int, base_lr: float, warmup_steps_cfg: int, total_steps: int) -> float:
    """Learning rate used by optimizer step number ``step`` (0-based); see ``lr_sequence``."""
    return lr_sequence(step + 1, base_lr, warmup_steps_cfg, total_steps)[step]


def train_step(st: SAEState, x: np.ndarray, lr: float, mode: str = "fp32", max_norm: float = 1.0,
               weight_decay: float = 0.0, beta1: float = 0.9, beta2: float = 0.999,
               eps: float = 1e-8, world_grads: list | None = None, select: np.ndarray | None = None,
               reduced_grads: dict | None = None) -> dict:
    """training.py:161-217 ``SAETrainer.train_step`` minus scheduler bookkeeping.

    forward (train mode) -> backward -> global-L2 clip -> AdamW -> decoder column renorm.
    ``world_grads``: optional list of gradient dicts from the other data-parallel ranks; they are
    averaged with this rank's before clipping (mean of per-rank mean-gradients).
    ``reduced_grads``: the already averaged gradients (replaces this rank's own: what came off the exchange buffer).
    """
    fwd = forward(st, x, mode, training=True, select=select)
    grads = backward(st, x, fwd, mode)
    if world_grads:
        n = len(world_grads) + 1
        for name in SAEState.PARAMS:
            acc = grads[name].astype(F64)
            for og in world_grads:
                acc = acc + og[name].astype(F64)
            grads[name] = (acc / n).astype(F32)
    if reduced_grads is not None:  # the data-parallel exchange has already happened (tests of the wire formats): use its result
        grads = {name: np.asarray(reduced_grads[name], dtype=F32) for name in SAEState.PARAMS}
    total = grad_total_norm(grads)
    coef = clip_coef(total, max_norm)
    st.adam_t += 1
    for name in SAEState.PARAMS:
        p = getattr(st, name)
        gcl = (grads[name].astype(F64) * coef).astype(F32)
        m = st.adam_m.get(name, np.zeros_like(p))
        v = st.adam_v.get(name, np.zeros_like(p))
        p, m, v = adamw_update(p, gcl, m, v, st.adam_t, lr, beta1, beta2, eps, weight_decay)
        setattr(st, name, p)
        st.adam_m[name], st.adam_v[name] = m, v
    st.W_d = normalize_decoder(st.W_d)  # training.py:197-198 (TopKSAE always normalises)
    return {"loss": float(fwd["loss"]), "l0": float(fwd["l0"]), "grad_norm": total, "clip_coef": coef,
            "dead_feature_ratio": dead_ratio(st), "fwd": fwd, "grads": grads}


# --------------------------------------------------------------------------------------------
# dead-feature resampling  (model.py:197-257)
# --------------------------------------------------------------------------------------------
def resample_dead_features(st: SAEState, inputs: np.ndarray, num_resample: int | None = None,
                           mode: str = "fp32", training: bool = True) -> dict:
    """model.py:197-257, quirks included.

    * dead indices ascending, capped to ``num_resample``;
    * ``self.forward(inputs)`` runs under ``no_grad`` but still bumps ``step_count`` /
      ``last_activated`` when the module is in train mode (model.py:229 -> :157);
    * per-row error = ``sum((inputs - recon)^2, -1)``; rows by error descending, ``min(num_dead, B)``;
    * direction = L2-normalised *raw input row* (model.py:237-240), written to ``W_e[idx, :]`` and
      ``W_d[:, idx]``, ``b_e[idx] = 0``, ``last_activated[idx] = step_count``;
    * Adam moments untouched; the return value is the capped dead count even when fewer rows than
      that were available (model.py:257).
    """
    inputs = np.asarray(inputs, dtype=F32)
    dead = np.nonzero(dead_mask(st))[0]
    num_dead = len(dead)
    if num_dead == 0:
        return {"returned": 0, "rewritten": np.zeros(0, dtype=np.int64), "rows": np.zeros(0, dtype=np.int64)}
    if num_resample is not None:
        num_dead = min(num_dead, num_resample)
        dead = dead[:num_dead]
    fwd = forward(st, inputs, mode, training=training)
    resid = inputs.astype(F64) - fwd["reconstructed"].astype(F64)
    errors = (resid * resid).sum(axis=1).astype(F32)
    n_take = min(num_dead, len(errors))
    rows = np.argsort(-errors.astype(F64), kind="stable")[:n_take]
    hi = inputs[rows].astype(F64)
    hi = (hi / np.maximum(np.sqrt((hi * hi).sum(axis=1, keepdims=True)), 1e-12)).astype(F32)
    n_write = min(num_dead, n_take)
    for i in range(n_write):
        f = dead[i]
        st.W_e[f, :] = hi[i]
        st.b_e[f] = 0.0
        st.W_d[:, f] = hi[i]
        st.last_activated[f] = st.step_count
    return {"returned": num_dead, "rewritten": dead[:n_write].copy(), "rows": rows, "errors": errors}


# --------------------------------------------------------------------------------------------
# ReLU SAE  (model.py:260-322)
# --------------------------------------------------------------------------------------------
def fp8_e4m3_round(a: np.ndarray) -> np.ndarray:
    """Nearest OCP e4m3fn value, ties to even (what v_cvt_pk_fp8_f32 does on gfx950: profiles/tools/probe_fp8.hip).
    3 mantissa bits, exponents 2^-6 .. 2^8, subnormal spacing 2^-9, largest finite 448; callers scale into range."""
    a = np.asarray(a, dtype=F64)
    mag = np.abs(a)
    e = np.floor(np.log2(np.maximum(mag, 2.0 ** -20)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    q = np.rint(mag / step) * step  # np.rint rounds half to even
    assert (q <= 448).all(), "fp8_e4m3_round: value outside the finite range"
    return (np.sign(a) * q).astype(F32)


def fp8_quant_rows(v: np.ndarray):
    """Per-row e4m3 quantisation as wsae_relu.hip's quant_rows_kernel: q = e4m3(v * (448 / amax)), scale = amax / 448,
    all in fp32 arithmetic.  Returns (q as float32 values, scale [rows])."""
    v = np.asarray(v, dtype=F32)
    amax = np.abs(v).max(axis=1).astype(F32)
    inv = np.where(amax > 0, F32(448.0) / np.where(amax > 0, amax, F32(1)), F32(1)).astype(F32)
    scale = np.where(amax > 0, amax / F32(448.0), F32(1)).astype(F32)
    return fp8_e4m3_round((v * inv[:, None]).astype(F32)), scale


def fp8_quant_tensor(v: np.ndarray):
    """Per-TENSOR e4m3 quantisation (one scale for the whole matrix, returned per row for the caller's convenience) as
    wsae_relu.hip quantises the decoder weight: q = e4m3(v * (448 / amax)), scale = amax / 448, fp32 arithmetic.  The
    decoder's columns are kept at unit norm, so its entries share one dynamic range; e4m3 is a floating-point format, the
    relative precision of an entry does not depend on where in the range it sits."""
    v = np.asarray(v, dtype=F32)
    amax = F32(np.abs(v).max())
    inv = F32(448.0) / amax if amax > 0 else F32(1)
    scale = amax / F32(448.0) if amax > 0 else F32(1)
    return fp8_e4m3_round((v * inv).astype(F32)), np.full(v.shape[0], scale, dtype=F32)


def relu_forward(W_e, b_e, W_d, b_d, x, sparsity_weight: float = 0.01, mode: str = "fp32") -> dict:
    """model.py:304-322: hidden = relu(enc(x)); recon = dec(hidden); loss = mse + w * mean|hidden|.

    ``mode="fp8"`` mirrors the device's fp8 forward (BASELINE.json configs[4]; wsae_ctx_set_relu_fp8): both GEMMs on
    e4m3 copies of their operands - x and bf16(hidden) per batch row, bf16(W_e) per feature row, bf16(W_d) with ONE scale
    for the matrix - with fp32 accumulation and ``row scale * column scale`` applied to the accumulator before the bias."""
    x = np.asarray(x, dtype=F32)
    if mode == "fp8":
        xq, sx = fp8_quant_rows(x)
        wq, sw = fp8_quant_rows(bf16_round(W_e))
        pre = ((xq.astype(F64) @ wq.astype(F64).T).astype(F32) * (sx[:, None] * sw[None, :]).astype(F32) + b_e).astype(F32)
        hidden = np.maximum(pre, 0).astype(F32)
        hq, sh = fp8_quant_rows(bf16_round(hidden))
        dq, sd = fp8_quant_tensor(bf16_round(W_d))
        recon = ((hq.astype(F64) @ dq.astype(F64).T).astype(F32) * (sh[:, None] * sd[None, :]).astype(F32) + b_d).astype(F32)
    else:
        pre = (x.astype(F64) @ W_e.astype(F64).T + b_e.astype(F64)).astype(F32)
        hidden = np.maximum(pre, 0).astype(F32)
        recon = (hidden.astype(F64) @ W_d.astype(F64).T + b_d.astype(F64)).astype(F32)
    resid = recon.astype(F64) - x.astype(F64)
    mse = F32(np.mean(resid * resid))
    l1 = F32(np.mean(np.abs(hidden.astype(F64))))
    loss = F32(F64(mse) + sparsity_weight * F64(l1))
    l0 = F32((hidden > 0).sum(axis=1).astype(F64).mean())
    return {"pre": pre, "hidden": hidden, "reconstructed": recon, "loss": loss,
            "reconstruction_loss": mse, "sparsity_loss": l1, "l0": l0}


def relu_backward(W_e, b_e, W_d, b_d, x, fwd: dict, sparsity_weight: float = 0.01) -> dict:
    """Autograd of model.py:304-311 (no ``b_pre``, no ``dx``)."""
    x = np.asarray(x, dtype=F32)
    B, D = x.shape
    H = W_e.shape[0]
    hidden = fwd["hidden"].astype(F64)
    g = 2.0 * (fwd["reconstructed"].astype(F64) - x.astype(F64)) / (B * D)
    dW_d = g.T @ hidden
    db_d = g.sum(axis=0)
    dh = g @ W_d.astype(F64) + sparsity_weight / (B * H) * np.sign(hidden)
    dpre = np.where(fwd["pre"] > 0, dh, 0.0)
    dW_e = dpre.T @ x.astype(F64)
    db_e = dpre.sum(axis=0)
    return {"W_e": dW_e.astype(F32), "b_e": db_e.astype(F32), "W_d": dW_d.astype(F32),
            "b_d": db_d.astype(F32)}


# --------------------------------------------------------------------------------------------
# Transcoders  (sae/transcoder.py:32-422)
# --------------------------------------------------------------------------------------------
def transcoder_forward(W_e, b_e, W_d, b_d, k, x, target, mode: str = "fp32", skip_W=None, skip_b=None,
                       select: np.ndarray | None = None) -> dict:
    """transcoder.py:112-176 (TopKTranscoder) / :346-403 (SkipTranscoder when ``skip_W`` is given):
    hidden = scatter(relu(topk(encoder(x)))); predicted = decoder(hidden) [+ skip(x)]; loss = mse(predicted, target).
    ``"amp"`` mirrors the device's bf16 mode exactly as for the SAE (no pre-bias here); the skip path is fp32."""
    x, target = np.asarray(x, dtype=F32), np.asarray(target, dtype=F32)
    if mode == "amp":
        pre = (bf16_round(x).astype(F64) @ bf16_round(W_e).astype(F64).T + b_e.astype(F64)).astype(F32)
        w_d = bf16_round(W_d)
    else:
        pre = (x.astype(F64) @ W_e.astype(F64).T + b_e.astype(F64)).astype(F32)
        w_d = W_d
    if select is None:
        vals, idx = topk_select(pre, k)
    else:
        idx = np.asarray(select, dtype=np.int64)
        vals = np.take_along_axis(pre, idx, axis=1).astype(F32)
    hidden = densify(vals, idx, W_e.shape[0])
    sparse = (hidden.astype(F64) @ w_d.astype(F64).T + b_d.astype(F64)).astype(F32)
    skip = None
    if skip_W is not None:
        skip = (x.astype(F64) @ skip_W.astype(F64).T + skip_b.astype(F64)).astype(F32)
    pred = sparse if skip is None else (sparse.astype(F64) + skip.astype(F64)).astype(F32)
    resid = pred.astype(F64) - target.astype(F64)
    loss = F32(np.mean(resid * resid))
    l0 = F32((hidden > 0).sum(axis=1).astype(F64).mean())
    return {"pre": pre, "vals": vals, "idx": idx, "hidden": hidden, "predicted": pred, "loss": loss, "l0": l0}


def transcoder_backward(W_e, b_e, W_d, b_d, x, target, fwd: dict, mode: str = "fp32", skip_W=None) -> dict:
    """Autograd of the above: g = 2 (predicted - target) / (B out); dW_d = g^T hidden; db_d = sum g;
    dpre = (g W_d) * 1[hidden > 0]; dW_e = dpre^T x; db_e = sum dpre; dx = dpre W_e [+ g W_skip];
    dW_skip = g^T x; db_skip = sum g."""
    x, target = np.asarray(x, dtype=F32), np.asarray(target, dtype=F32)
    B, Dout = target.shape
    hidden = fwd["hidden"].astype(F64)
    g = (2.0 * (fwd["predicted"].astype(F64) - target.astype(F64)) / (B * Dout)).astype(F32)
    g64 = g.astype(F64)
    if mode == "amp":
        dh = bf16_round(g).astype(F64) @ bf16_round(W_d).astype(F64)
        dpre = bf16_round(np.where(hidden > 0, dh, 0.0).astype(F32)).astype(F64)
        dW_d = bf16_round(g).astype(F64).T @ bf16_round(hidden.astype(F32)).astype(F64)
        dW_e = dpre.T @ bf16_round(x).astype(F64)
        dx = dpre @ W_e.astype(F64)
    else:
        dh = g64 @ W_d.astype(F64)
        dpre = np.where(hidden > 0, dh, 0.0)
        dW_d = g64.T @ hidden
        dW_e = dpre.T @ x.astype(F64)
        dx = dpre @ W_e.astype(F64)
    out = {"W_e": dW_e.astype(F32), "b_e": dpre.sum(axis=0).astype(F32), "W_d": dW_d.astype(F32),
           "b_d": g64.sum(axis=0).astype(F32), "g": g}
    if skip_W is not None:
        out["skip_W"] = (g64.T @ x.astype(F64)).astype(F32)
        out["skip_b"] = g64.sum(axis=0).astype(F32)
        dx = dx + g64 @ skip_W.astype(F64)
    out["x"] = dx.astype(F32)
    return out


# --------------------------------------------------------------------------------------------
# Cross-layer crosscoder, TopK variant  (sae/crosscoder.py:286-379 on top of :38-283)
# --------------------------------------------------------------------------------------------
def crosscoder_forward(W_enc, b_enc, W_dec, b_dec, k, layer_acts, mode: str = "fp32", select=None) -> dict:
    """crosscoder.py:323-379.  ``W_enc [L, d, S]``, ``W_dec [S, L, d]``, ``b_dec [L, d]``; ``layer_acts`` = list of the
    L per-layer ``[B, d]`` inputs (internal layer order).  pre = sum_l acts_l @ W_enc[l] + b_enc (:331-338); hidden =
    scatter(relu(topk(pre))) (:341-343); recon_l = hidden @ W_dec[:, l, :] + b_dec[l] (:181-184); loss = sum_l
    mean((recon_l - acts_l)^2) (:352-358).  ``"amp"`` rounds the GEMM operands to bf16 as the device's bf16 mode does."""
    L, d, S = W_enc.shape
    acts = [np.asarray(a, dtype=F32) for a in layer_acts]
    rd = bf16_round if mode == "amp" else (lambda a: a)
    pre64 = np.zeros((acts[0].shape[0], S), dtype=F64)
    for l in range(L):
        pre64 += rd(acts[l]).astype(F64) @ rd(W_enc[l]).astype(F64)
    pre = (pre64 + b_enc.astype(F64)).astype(F32)
    if select is None:
        vals, idx = topk_select(pre, k)
    else:
        idx = np.asarray(select, dtype=np.int64)
        vals = np.take_along_axis(pre, idx, axis=1).astype(F32)
    hidden = densify(vals, idx, S)
    recon, per_layer = [], []
    for l in range(L):
        r = (hidden.astype(F64) @ rd(W_dec[:, l, :]).astype(F64) + b_dec[l].astype(F64)).astype(F32)
        recon.append(r)
        e = r.astype(F64) - acts[l].astype(F64)
        per_layer.append(F32(np.mean(e * e)))
    loss = F32(np.sum(np.asarray(per_layer, dtype=F64)))
    l0 = F32((hidden > 0).sum(axis=1).astype(F64).mean())
    return {"pre": pre, "vals": vals, "idx": idx, "hidden": hidden, "recon": recon, "per_layer_loss": per_layer,
            "loss": loss, "l0": l0}


def crosscoder_backward(W_enc, b_enc, W_dec, b_dec, layer_acts, fwd: dict) -> dict:
    """Autograd of ``crosscoder_forward``'s loss (fp32 mode): g_l = 2 (recon_l - acts_l) / (B d); dW_dec[:, l, :] =
    hidden^T g_l; db_dec[l] = sum g_l; dpre = (sum_l g_l W_dec[:, l, :]^T) * 1[hidden > 0]; dW_enc[l] = acts_l^T dpre;
    db_enc = sum dpre."""
    L, d, S = W_enc.shape
    acts = [np.asarray(a, dtype=F64) for a in layer_acts]
    B = acts[0].shape[0]
    hidden = fwd["hidden"].astype(F64)
    g = [2.0 * (fwd["recon"][l].astype(F64) - acts[l]) / (B * d) for l in range(L)]
    dh = np.zeros((B, S), dtype=F64)
    dW_dec = np.zeros(W_dec.shape, dtype=F64)
    for l in range(L):
        dh += g[l] @ W_dec[:, l, :].astype(F64).T
        dW_dec[:, l, :] = hidden.T @ g[l]
    dpre = np.where(hidden > 0, dh, 0.0)
    dW_enc = np.stack([acts[l].T @ dpre for l in range(L)])
    return {"W_enc": dW_enc.astype(F32), "b_enc": dpre.sum(axis=0).astype(F32), "W_dec": dW_dec.astype(F32),
            "b_dec": np.stack([g[l].sum(axis=0) for l in range(L)]).astype(F32)}


def crosscoder_relu_forward(W_enc, b_enc, W_dec, b_dec, layer_acts, sparsity_weight: float) -> dict:
    """CrossLayerCrosscoder with activation "relu" (crosscoder.py:142-235): hidden = relu(sum_l acts_l @ W_enc[l] + b_enc);
    loss = sum_l mean((recon_l - acts_l)^2) + sparsity_weight * mean_b(sum_s |h_bs| n_s), n_s = ||W_dec[s].flatten()||."""
    L, d, S = W_enc.shape
    acts = [np.asarray(a, dtype=F32) for a in layer_acts]
    pre = np.zeros((acts[0].shape[0], S), dtype=F64)
    for l in range(L):
        pre += acts[l].astype(F64) @ W_enc[l].astype(F64)
    hidden = np.maximum((pre + b_enc.astype(F64)).astype(F32), F32(0))
    recon, per_layer = [], []
    for l in range(L):
        r = (hidden.astype(F64) @ W_dec[:, l, :].astype(F64) + b_dec[l].astype(F64)).astype(F32)
        recon.append(r)
        e = r.astype(F64) - acts[l].astype(F64)
        per_layer.append(F32(np.mean(e * e)))
    norms = np.sqrt((W_dec.astype(F64).reshape(S, -1) ** 2).sum(axis=1))
    sparsity = F32(np.mean(np.abs(hidden).astype(F64) @ norms))
    recon_loss = F32(np.sum(np.asarray(per_layer, dtype=F64)))
    return {"hidden": hidden, "recon": recon, "per_layer_loss": per_layer, "reconstruction_loss": recon_loss,
            "sparsity_loss": sparsity, "loss": F32(F64(recon_loss) + sparsity_weight * F64(sparsity)),
            "l0": F32((hidden > 0).sum(axis=1).astype(F64).mean()), "norms": norms.astype(F32)}


def crosscoder_relu_backward(W_enc, b_enc, W_dec, b_dec, layer_acts, fwd: dict, sparsity_weight: float) -> dict:
    """Autograd of the above: g_l = 2 r_l / (B d); dpre = (sum_l g_l W_dec[:, l]^T + sparsity_weight n_s / B) 1[h > 0];
    dW_dec[s, l] = h_s^T g_l + sparsity_weight mean_b|h_bs| W_dec[s, l] / n_s (the norm's own gradient)."""
    L, d, S = W_enc.shape
    acts = [np.asarray(a, dtype=F64) for a in layer_acts]
    B = acts[0].shape[0]
    hidden = fwd["hidden"].astype(F64)
    norms = fwd["norms"].astype(F64)
    g = [2.0 * (fwd["recon"][l].astype(F64) - acts[l]) / (B * d) for l in range(L)]
    dh = np.zeros((B, S), dtype=F64) + sparsity_weight * norms[None, :] / B
    dW_dec = np.zeros(W_dec.shape, dtype=F64)
    for l in range(L):
        dh += g[l] @ W_dec[:, l, :].astype(F64).T
        dW_dec[:, l, :] = hidden.T @ g[l]
    dW_dec += (sparsity_weight * np.abs(hidden).mean(axis=0) / norms)[:, None, None] * W_dec.astype(F64)
    dpre = np.where(hidden > 0, dh, 0.0)
    return {"W_enc": np.stack([acts[l].T @ dpre for l in range(L)]).astype(F32), "b_enc": dpre.sum(axis=0).astype(F32),
            "W_dec": dW_dec.astype(F32), "b_dec": np.stack([g[l].sum(axis=0) for l in range(L)]).astype(F32)}
