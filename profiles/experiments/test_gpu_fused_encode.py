"""The fused encoder + TopK filter (wsae_encode_fused.hip) against the dense path it replaces and the oracle.

The filter kernel accumulates every pre-activation exactly as the dense GEMM does (same MFMA, same K order, bias
last), so (values, indices) must be BIT-IDENTICAL to TopK over ``wsae_encode_dense`` of the same batch -- at the
benchmark's sizes, for ragged batches, for every part count the launcher picks, and for rows that take the
fallback (threshold too high / list overflow).  reference: model.py:108-116."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O
from oracle import synth
from whisper_sae import _native as N
from whisper_sae.sae.model import TopKSAE

pytestmark = pytest.mark.gpu


def make(D, H, K, device, seed=3, bump=None):
    torch.manual_seed(seed)
    m = TopKSAE(D, H, k=K).to(device)
    m.precision = "bf16"
    with torch.no_grad():
        m.b_pre.normal_(0, 0.05)
        m.encoder.bias.normal_(0, 0.1)
        if bump is not None:
            m.encoder.bias[bump] += 25.0
    return m


def dense_topk(m, x):
    pre = m.pre_activation(x)
    v, i = torch.topk(pre, m.k, dim=-1)
    return v, i.to(torch.int32)


def code_sorted(v, i):
    """canonical order (value descending, index ascending) for comparison"""
    key = torch.argsort(i, dim=1, stable=True)
    v, i = torch.gather(v, 1, key), torch.gather(i, 1, key)
    key = torch.argsort(v, dim=1, descending=True, stable=True)
    return torch.gather(v, 1, key), torch.gather(i, 1, key)


def fallback_rows(m):
    """cumulative count of rows that took the exact TopK path on this model's engine"""
    return int(m.bind().stats[6].item())


def path(m, x):
    eng = m.bind()
    h = eng.prepare(N.PREC_BF16, x.shape[0], force=True)
    return N.lib().wsae_ctx_encode_path(h, N.DT_BF16 if x.dtype == torch.bfloat16 else N.DT_F32, x.shape[0])


@pytest.mark.parametrize("D,H,K,B", [(384, 3072, 32, 16384), (384, 3072, 32, 4096), (384, 3072, 32, 1500),
                                     (256, 2048, 16, 2048), (128, 4096, 64, 1024), (384, 3072, 8, 1027)])
def test_bit_identical_to_dense_topk(device, D, H, K, B):
    m = make(D, H, K, device)
    x = torch.randn(B, D, device=device).to(torch.bfloat16)
    assert path(m, x) == 1
    fb0 = fallback_rows(m)
    v, i = m.encode_compact(x)
    fb = fallback_rows(m) - fb0
    vd, idd = dense_topk(m, x)
    v, i = code_sorted(v, i)
    vd, idd = code_sorted(vd, idd)
    assert torch.equal(i, idd)
    if fb == 0:
        assert torch.equal(v, vd)
    else:  # a row on the exact path sums in another order
        assert torch.allclose(v, vd, rtol=2e-6, atol=2e-6) and (v != vd).any(dim=1).sum() <= fb
    # the threshold is meant to settle nearly every row: a handful per batch may take the exact path
    assert fb <= max(8, B // 500), fb


def test_small_batches_and_fp32_rows_keep_the_dense_path(device):
    m = make(384, 3072, 32, device)
    assert path(m, torch.zeros(512, 384, device=device, dtype=torch.bfloat16)) == 0
    assert path(m, torch.zeros(4096, 384, device=device)) == 0


def test_rows_the_filter_cannot_settle_take_the_exact_path(device):
    # 128 features with a huge bias all sit in the first part's calibration slabs: that part's threshold ends up far
    # above what the other parts file, fewer than k candidates clear the largest threshold, every row falls back
    D, H, K, B = 384, 3072, 32, 1024
    m = make(D, H, K, device, bump=slice(0, 128))
    x = torch.randn(B, D, device=device).to(torch.bfloat16)
    assert path(m, x) == 1
    fb0 = fallback_rows(m)
    v, i = m.encode_compact(x)
    assert fallback_rows(m) - fb0 > B // 2  # (nearly every row; a row can still clear the bumped threshold by chance)
    vd, idd = dense_topk(m, x)
    v, i = code_sorted(v, i)
    vd, idd = code_sorted(vd, idd)
    assert torch.equal(i, idd)
    # the fallback accumulates with plain FMAs (another summation order than the MFMA): values agree to rounding
    assert torch.allclose(v, vd, rtol=2e-6, atol=2e-6)
    # and the next batch starts from a clean fallback list
    with torch.no_grad():
        m.encoder.bias[:128] -= 25.0
    fb0 = fallback_rows(m)
    v2, i2 = m.encode_compact(x)
    assert fallback_rows(m) - fb0 <= 8
    vd, idd = dense_topk(m, x)
    assert torch.equal(code_sorted(v2, i2)[1], code_sorted(vd, idd)[1])


def test_mixed_rows_some_fall_back(device):
    # a bias bump on a few hundred scattered features makes the calibration of SOME rows/parts unrepresentative
    D, H, K, B = 384, 3072, 32, 4096
    g = torch.Generator().manual_seed(11)
    bump = torch.randperm(H, generator=g)[:40]
    m = make(D, H, K, device, bump=bump)
    x = torch.randn(B, D, device=device).to(torch.bfloat16)
    v, i = m.encode_compact(x)
    vd, idd = dense_topk(m, x)
    v, i = code_sorted(v, i)
    vd, idd = code_sorted(vd, idd)
    assert torch.equal(i, idd)
    assert torch.allclose(v, vd, rtol=2e-6, atol=2e-6)


def test_against_the_amp_oracle(device):
    D, H, K, B = 384, 3072, 32, 2048
    w = synth.sae_weights(D, H, seed=42, bf16=True, b_pre_scale=0.1)
    x = synth.activations(B, D, seed=42, stream=1, bf16=True)
    m = TopKSAE(D, H, k=K).to(device)
    m.precision = "bf16"
    m.load_state_dict({k_: torch.from_numpy(np.array(v)) for k_, v in w.items()}, strict=False)
    xb = torch.from_numpy(x).to(device).to(torch.bfloat16)
    assert path(m, xb) == 1
    v, i = m.encode_compact(xb)
    f = O.forward(O.SAEState.from_state_dict(w, k=K), x, "amp", training=False)
    got = np.sort(i.cpu().numpy(), axis=1)
    want = np.sort(f["idx"], axis=1)
    same = (got == want).all(axis=1)
    # rows that differ must be near-ties of the oracle's own pre-activations
    if not same.all():
        assert O.check_selection(f["pre"][~same], got[~same], K, rtol=1e-5).all()
    assert same.mean() > 0.99
