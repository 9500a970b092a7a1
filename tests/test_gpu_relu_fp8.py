"""The ReLU SAE's fp8 forward (BASELINE.json configs[4]: "fp8 MFMA encode/decode"; reference model.py:304-322).

``ReLUSAE(precision="fp8")`` = bf16 mode with both forward GEMMs on OCP e4m3 operands (``wsae_ctx_set_relu_fp8``):
per-row scales, ``v_mfma_f32_32x32x16_fp8_fp8``, fp32 accumulate.  Held to the oracle's ``"fp8"`` mode, which mirrors
the quantisation (its e4m3 rounding equals ``torch.float8_e4m3fn``'s: tests/test_oracle_golden.py) - and bounded
against the reference's fp32 values, with the tolerance fp8 operands leave."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O
from oracle import synth
from whisper_sae import _native as N
from whisper_sae.sae.model import ReLUSAE

pytestmark = pytest.mark.gpu


def make(D, H, seed, weight=0.01, precision="fp8", device="cuda:0"):
    w = synth.sae_weights(D, H, seed=seed, bf16=True)
    m = ReLUSAE(D, H, sparsity_weight=weight, precision=precision)
    sd = m.state_dict()
    for key in ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias"):
        sd[key] = torch.from_numpy(w[key])
    m.load_state_dict(sd)
    return m.to(device), (w["encoder.weight"], w["encoder.bias"], w["decoder.weight"], w["decoder.bias"])


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("D,H,B", [(256, 1024, 1024), (512, 2048, 700), (1280, 4096, 512)])
def test_forward_matches_the_fp8_oracle(device, D, H, B):
    m, W = make(D, H, seed=31, device=device)
    x = synth.activations(B, D, seed=31, stream=3, bf16=True)
    with torch.no_grad():
        out = m(torch.from_numpy(x).to(device))
    f = O.relu_forward(*W, x, sparsity_weight=0.01, mode="fp8")
    # the encoder GEMM sees bit-identical e4m3 operands on both sides: hidden differs by summation order only
    assert rel(out.hidden.cpu().numpy(), f["hidden"]) < 5e-5  # (fp32 accumulation of up to 1280 products of magnitude <= 448^2)
    assert abs(float(out.l0) - float(f["l0"])) <= 1e-3 * float(f["l0"])
    # the decoder GEMM quantises bf16(hidden): a hidden value that rounds the other way moves one operand by an e4m3 step
    assert rel(out.reconstructed.cpu().numpy(), f["reconstructed"]) < 8e-3
    assert abs(float(out.loss) - float(f["loss"])) < 1e-4 * float(f["loss"])
    # against the reference's fp32 values: what 3 mantissa bits per operand leave
    g = O.relu_forward(*W, x, sparsity_weight=0.01)
    assert abs(float(out.loss) - float(g["loss"])) < 2e-2 * float(g["loss"])
    r = out.reconstructed.cpu().numpy().astype(np.float64) - g["reconstructed"]
    assert np.linalg.norm(r) / np.linalg.norm(g["reconstructed"]) < 6e-2


def test_training_in_fp8_tracks_bf16(device, tmp_path):
    from whisper_sae.config import TrainingConfig
    from whisper_sae.sae.training import SAETrainer
    D, H, B = 256, 2048, 1024
    losses = {}
    for prec in ("bf16", "fp8"):
        m, _ = make(D, H, seed=5, precision=prec, device=device)
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.0, epochs=1, warmup_steps=0, gradient_clip=1.0,
                             use_amp=True, num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path / prec)
        seq = []
        for s in range(12):
            x = synth.activations(B, D, seed=5, stream=10 + s, bf16=True)
            seq.append(tr.train_step(torch.from_numpy(x).to(device)).loss)
        losses[prec] = np.array(seq)
        cn = m.decoder.weight.detach().norm(dim=0)
        assert torch.allclose(cn, torch.ones_like(cn), atol=1e-5)
    # the fp8 forward only perturbs the trajectory the bf16 mode takes (loss within 3 % at every step, same direction)
    assert np.abs(losses["fp8"] - losses["bf16"]).max() < 3e-2 * losses["bf16"].max()
    assert np.sign(losses["fp8"][-1] - losses["fp8"][0]) == np.sign(losses["bf16"][-1] - losses["bf16"][0])


def test_small_and_tail_batches_run_on_the_bf16_path(device, tmp_path):
    """A batch the persistent fp8 GEMM cannot take (< 512 rows: the tail batch of an epoch, an eval call) is computed on the
    bf16 path - decided from the shape before anything is queued - and equals the bf16 module bit for bit."""
    from whisper_sae.config import TrainingConfig
    from whisper_sae.sae.training import SAETrainer
    m8, _ = make(256, 1024, seed=2, device=device)
    mb, _ = make(256, 1024, seed=2, precision="bf16", device=device)
    x = torch.from_numpy(synth.activations(64, 256, seed=2, stream=1, bf16=True)).to(device)
    with torch.no_grad():
        a, b = m8(x), mb(x)
    assert torch.equal(a.hidden, b.hidden) and torch.equal(a.reconstructed, b.reconstructed) and float(a.loss) == float(b.loss)
    # an epoch whose last batch is short: 1024, 1024, 300 rows
    cfg = TrainingConfig(batch_size=1024, learning_rate=1e-3, warmup_steps=0, use_amp=True, num_workers=0)
    tr = SAETrainer(m8, cfg, device=device, run_dir=tmp_path / "tail")
    for n, s in ((1024, 3), (1024, 4), (300, 5)):
        met = tr.train_step(torch.from_numpy(synth.activations(n, 256, seed=2, stream=s, bf16=True)).to(device))
        assert np.isfinite(met.loss)
    # the fp8 forward still belongs to the bf16 mode
    tr32 = SAETrainer(m8, TrainingConfig(batch_size=1024, use_amp=False, num_workers=0), device=device, run_dir=tmp_path)
    with pytest.raises(N.WsaeError):
        tr32.train_step(torch.zeros(1024, 256, device=device))


def test_configs4_dimensions(device):
    # 1280 -> 40960: one forward on the fp8 GEMMs against the mirrored oracle
    D, H, B = 1280, 40960, 512
    m, W = make(D, H, seed=17, device=device)
    x = synth.activations(B, D, seed=17, stream=2, bf16=True)
    with torch.no_grad():
        out = m(torch.from_numpy(x).to(device))
    f = O.relu_forward(*W, x, sparsity_weight=0.01, mode="fp8")
    assert rel(out.hidden.cpu().numpy(), f["hidden"]) < 5e-5  # (fp32 accumulation of up to 1280 products of magnitude <= 448^2)
    assert abs(float(out.loss) - float(f["loss"])) < 2e-4 * float(f["loss"])
    assert rel(out.reconstructed.cpu().numpy(), f["reconstructed"]) < 8e-3
