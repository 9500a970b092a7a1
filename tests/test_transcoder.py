"""Transcoders (reference src/whisper_sae/sae/transcoder.py, tests/test_transcoder.py).

CPU: the oracle's transcoder restatement against golden set G12 (forward, every gradient, the skip path) produced by
the real reference.  GPU: the drop-in modules (TopK-SAE kernels, target != input, optional padding and skip path)
against G12 and the oracle, plus the reference's own behavioural tests restated under their original names."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O
from oracle import synth


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def case(g, tag):
    Din, Dout, H, K, B = (int(v) for v in g[f"{tag}.dims"])
    w = synth.sae_weights(Din, H, seed=23, bf16=False)
    x = synth.activations(B, Din, seed=23, stream=6, bf16=False)
    tgt = synth.activations(B, Dout, seed=23, stream=7, bf16=False)
    return (Din, Dout, H, K, B), (w["encoder.weight"], w["encoder.bias"], g[f"{tag}.W_d"], g[f"{tag}.b_d"]), x, tgt


@pytest.fixture(scope="module")
def g12(golden_dir):
    return np.load(golden_dir / "g12_transcoders.npz")


class TestOracleAgainstReference:
    @pytest.mark.parametrize("tag", ["eq", "narrow"])
    def test_topk_transcoder(self, g12, tag):
        (Din, Dout, H, K, B), W, x, tgt = case(g12, tag)
        f = O.transcoder_forward(*W, K, x, tgt)
        assert np.array_equal(np.sort(f["idx"], axis=1).astype(np.int16), g12[f"{tag}.idx"])
        assert rel(f["predicted"], g12[f"{tag}.pred"]) < 1e-5
        assert abs(float(f["loss"]) - float(g12[f"{tag}.loss"])) / float(g12[f"{tag}.loss"]) < 1e-5
        assert float(f["l0"]) == float(g12[f"{tag}.l0"])
        b = O.transcoder_backward(*W, x, tgt, f)
        for n, key in (("W_e", "dW_e"), ("b_e", "db_e"), ("W_d", "dW_d"), ("b_d", "db_d"), ("x", "dx")):
            assert rel(b[n], g12[f"{tag}.{key}"]) < 2e-5, n

    def test_skip_transcoder(self, g12):
        (Din, Dout, H, K, B), W, x, tgt = case(g12, "eq")
        ws, bs = g12["skip.W_s"], g12["skip.b_s"]
        f = O.transcoder_forward(*W, K, x, tgt, skip_W=ws, skip_b=bs)
        assert rel(f["predicted"], g12["skip.pred"]) < 1e-5
        assert abs(float(f["loss"]) - float(g12["skip.loss"])) / float(g12["skip.loss"]) < 1e-5
        b = O.transcoder_backward(*W, x, tgt, f, skip_W=ws)
        for n, key in (("W_e", "dW_e"), ("W_d", "dW_d"), ("b_d", "db_d"), ("skip_W", "dW_s"), ("skip_b", "db_s"), ("x", "dx")):
            assert rel(b[n], g12[f"skip.{key}"]) < 2e-5, n


class TestHostSurface:
    def test_state_dict_keys_and_factory(self):
        from whisper_sae.sae.transcoder import SkipTranscoder, TopKTranscoder, TranscoderOutput, create_transcoder
        t = TopKTranscoder(64, 32, 128, k=8)
        assert list(t.state_dict().keys()) == ["feature_last_activated", "step_count", "encoder.weight", "encoder.bias",
                                               "decoder.weight", "decoder.bias"]
        assert t.decoder.weight.shape == (32, 128) and t.encoder.weight.shape == (128, 64)
        cn = t.decoder.weight.detach().norm(dim=0)
        assert torch.allclose(cn, torch.full_like(cn, 0.1), atol=1e-6)   # unit-norm columns x 0.1 (ref :96-103)
        s = SkipTranscoder(64, 64, 128, k=8)
        assert [n for n, _ in s.named_parameters()] == ["encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias",
                                                        "skip.weight", "skip.bias"]
        for p in (s.decoder.weight, s.decoder.bias, s.skip.weight, s.skip.bias):   # ref test_paper_initialization_zeros
            assert torch.all(p == 0)
        assert not hasattr(s, "resample_dead_features") and hasattr(t, "resample_dead_features")
        assert isinstance(create_transcoder(64, 64, 128, k=16, use_skip=True), SkipTranscoder)
        assert isinstance(create_transcoder(64, 64, 128, k=16, use_skip=False), TopKTranscoder)
        assert create_transcoder(64, 64, 128, k=16, use_skip=False, dead_feature_threshold=500).dead_feature_threshold == 500
        assert TranscoderOutput._fields == ("predicted", "hidden", "loss", "reconstruction_loss", "sparsity_loss", "l0")

    def test_no_cpu_path(self):
        from whisper_sae import _native as N
        from whisper_sae.sae.transcoder import TopKTranscoder
        with pytest.raises(N.WsaeError):
            TopKTranscoder(64, 64, 128, k=8)(torch.randn(4, 64), torch.randn(4, 64))


def load(mod, W):
    sd = mod.state_dict()
    for key, val in zip(("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias"), W):
        sd[key] = torch.from_numpy(np.array(val))
    mod.load_state_dict(sd)


@pytest.mark.gpu
class TestDeviceAgainstReference:
    @pytest.mark.parametrize("tag", ["eq", "narrow"])
    def test_topk_transcoder_fp32(self, g12, device, tag):
        from whisper_sae.sae.transcoder import TopKTranscoder
        (Din, Dout, H, K, B), W, x, tgt = case(g12, tag)
        m = TopKTranscoder(Din, Dout, H, k=K, dead_feature_threshold=20, precision="fp32")
        load(m, W)
        m.to(device).train()
        xt = torch.from_numpy(x).to(device).requires_grad_(True)
        o = m(xt, torch.from_numpy(tgt).to(device))
        o.loss.backward()
        assert o.predicted.shape == (B, Dout) and o.hidden.shape == (B, H)
        assert rel(o.predicted.cpu().numpy(), g12[f"{tag}.pred"]) < 1e-5
        assert abs(float(o.loss.detach()) - float(g12[f"{tag}.loss"])) / float(g12[f"{tag}.loss"]) < 1e-5
        assert float(o.l0) == float(g12[f"{tag}.l0"]) and float(o.sparsity_loss) == 0.0
        _, idx = m._last_code
        assert np.array_equal(np.sort(idx.cpu().numpy(), axis=1).astype(np.int16), g12[f"{tag}.idx"])
        got = {"dW_e": m.encoder.weight.grad, "db_e": m.encoder.bias.grad, "dW_d": m.decoder.weight.grad,
               "db_d": m.decoder.bias.grad, "dx": xt.grad}
        for key, t in got.items():
            assert t.shape == g12[f"{tag}.{key}"].shape
            assert rel(t.cpu().numpy(), g12[f"{tag}.{key}"]) < 2e-5, key
        assert int(m.step_count.item()) == int(g12[f"{tag}.step_count"])
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), g12[f"{tag}.last_activated"])

    def test_skip_transcoder_fp32(self, g12, device):
        from whisper_sae.sae.transcoder import SkipTranscoder
        (Din, Dout, H, K, B), W, x, tgt = case(g12, "eq")
        m = SkipTranscoder(Din, Dout, H, k=K, precision="fp32")
        load(m, W)
        with torch.no_grad():
            m.skip.weight.copy_(torch.from_numpy(g12["skip.W_s"]))
            m.skip.bias.copy_(torch.from_numpy(g12["skip.b_s"]))
        m.to(device).train()
        xt = torch.from_numpy(x).to(device).requires_grad_(True)
        tt = torch.from_numpy(tgt).to(device)
        o = m(xt, tt)
        o.loss.backward()
        assert rel(o.predicted.cpu().numpy(), g12["skip.pred"]) < 1e-5
        assert abs(float(o.loss.detach()) - float(g12["skip.loss"])) / float(g12["skip.loss"]) < 1e-5
        got = {"dW_e": m.encoder.weight.grad, "dW_d": m.decoder.weight.grad, "db_d": m.decoder.bias.grad,
               "dW_s": m.skip.weight.grad, "db_s": m.skip.bias.grad, "dx": xt.grad}
        for key, t in got.items():
            assert rel(t.cpu().numpy(), g12[f"skip.{key}"]) < 2e-5, key
        assert abs(m.get_skip_contribution(xt.detach(), tt) - float(g12["skip.contribution"])) < 1e-5

    def test_resample_dead_features(self, g12, device):
        from whisper_sae.sae.transcoder import TopKTranscoder
        (Din, Dout, H, K, B), W, x, tgt = case(g12, "eq")
        m = TopKTranscoder(Din, Dout, H, k=K, dead_feature_threshold=20, precision="fp32")
        load(m, W)
        m.to(device).train()
        with torch.no_grad():
            m.step_count.fill_(100)
            la = torch.full((H,), 95, dtype=torch.long)
            la[torch.from_numpy(g12["rs.dead_idx"])] = 3
            m.feature_last_activated.copy_(la.to(device))
        ret = m.resample_dead_features(torch.from_numpy(x).to(device), torch.from_numpy(tgt).to(device))
        assert ret == int(g12["rs.ret"])
        sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        assert int(sd["step_count"]) == int(g12["rs.step_count"])
        assert np.array_equal(sd["feature_last_activated"], g12["rs.last_activated"])
        assert rel(sd["encoder.weight"], g12["rs.W_e"]) < 1e-6
        assert rel(sd["decoder.weight"], g12["rs.W_d"]) < 1e-5
        assert np.array_equal(sd["encoder.bias"] == 0, g12["rs.b_e"] == 0)

    def test_bf16_at_whisper_tiny_width_against_amp_oracle(self, device):
        """384 -> 384 through 3072 features, k = 32, B = 2048, bf16 mode: the persistent GEMM gathering its rows, the
        strip-guided TopK, the MFMA decode kernel with a target that is not the input."""
        from whisper_sae.sae.transcoder import TopKTranscoder
        D, H, K, B = 384, 3072, 32, 2048
        w = synth.sae_weights(D, H, seed=29, bf16=False)
        W = (w["encoder.weight"], w["encoder.bias"], w["decoder.weight"], w["decoder.bias"])
        x = synth.activations(B, D, seed=29, stream=1, bf16=True)
        tgt = synth.activations(B, D, seed=29, stream=2, bf16=True)
        m = TopKTranscoder(D, D, H, k=K, precision="bf16")
        load(m, W)
        m.to(device).train()
        o = m(torch.from_numpy(x).to(device).to(torch.bfloat16), torch.from_numpy(tgt).to(device).to(torch.bfloat16))
        o.loss.backward()
        _, idx = m._last_code
        pre = O.transcoder_forward(*W, K, x, tgt, "amp")["pre"]
        clear = synth.topk_margin(pre, K) > 1e-5
        assert O.check_selection(pre, idx.cpu().numpy(), K).all()
        f = O.transcoder_forward(*W, K, x, tgt, "amp", select=idx.cpu().numpy())
        assert clear.mean() > 0.98
        assert abs(float(o.loss.detach()) - float(f["loss"])) / float(f["loss"]) < 1e-5
        assert rel(o.predicted.cpu().numpy(), f["predicted"]) < 1e-5
        b = O.transcoder_backward(*W, x, tgt, f, "amp")
        for n, t in (("W_e", m.encoder.weight.grad), ("b_e", m.encoder.bias.grad), ("W_d", m.decoder.weight.grad),
                     ("b_d", m.decoder.bias.grad)):
            assert rel(t.cpu().numpy(), b[n]) < 2e-3, n


@pytest.mark.gpu
class TestReferenceBehaviour:
    """tests/test_transcoder.py of the reference, restated on device tensors (names kept)."""

    def _mk(self, device, skip=False, **kw):
        from whisper_sae.sae.transcoder import create_transcoder
        torch.manual_seed(0)
        args = dict(input_dim=64, output_dim=64, hidden_dim=128, k=16, use_skip=skip)
        args.update(kw)
        return create_transcoder(**args).to(device)

    def test_topk_sparsity(self, device):
        m = self._mk(device)
        h = m.encode(torch.randn(32, 64, device=device))
        assert h.shape == (32, 128) and torch.all((h != 0).sum(dim=-1) <= 16) and torch.all(h >= 0)

    def test_different_input_output_dims(self, device):
        m = self._mk(device, output_dim=32)
        o = m(torch.randn(8, 64, device=device), torch.randn(8, 32, device=device))
        assert o.predicted.shape == (8, 32)
        assert m.decode(torch.randn(8, 128, device=device)).shape == (8, 32)

    def test_loss_is_mse(self, device):
        m = self._mk(device)
        x, y = torch.randn(32, 64, device=device), torch.randn(32, 64, device=device)
        o = m(x, y)
        assert torch.allclose(o.loss, torch.nn.functional.mse_loss(o.predicted, y), rtol=1e-5)
        assert torch.equal(o.loss, o.reconstruction_loss)

    def test_dead_feature_tracking(self, device):
        m = self._mk(device, dead_feature_threshold=5)
        m.train()
        assert m.get_dead_feature_ratio() == 0.0
        x, y = torch.randn(4, 64, device=device), torch.randn(4, 64, device=device)
        for _ in range(10):
            m(x, y)
        assert int(m.step_count.item()) == 10 and 0.0 < m.get_dead_feature_ratio() < 1.0

    def test_forward_includes_skip(self, device):
        m = self._mk(device, skip=True)
        with torch.no_grad():
            m.skip.weight.copy_(torch.eye(64))
        x = torch.randn(16, 64, device=device)
        o = m(x, torch.randn(16, 64, device=device))
        # decoder weights are zero at initialisation: the prediction is the skip path plus the decoder bias (zero)
        assert torch.allclose(o.predicted, x, atol=1e-5)

    def test_set_output_bias(self, device):
        m = self._mk(device, skip=True)
        mean = torch.randn(64, device=device)
        m.set_output_bias(mean)
        o = m(torch.randn(8, 64, device=device), torch.randn(8, 64, device=device))
        assert torch.allclose(o.predicted, mean.expand(8, 64), atol=1e-5)

    def test_loss_decreases_with_training(self, device):
        m = self._mk(device, hidden_dim=256, k=32)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        torch.manual_seed(1)
        w_true = torch.randn(64, 64, device=device) * 0.3
        x = torch.randn(512, 64, device=device)
        y = torch.tanh(x @ w_true)
        losses = []
        for _ in range(60):
            opt.zero_grad()
            o = m(x, y)
            o.loss.backward()
            opt.step()
            m.normalize_decoder_weights()
            losses.append(float(o.loss.detach()))
        assert losses[-1] < 0.7 * losses[0]

    def test_skip_helps_linear_transformations(self, device):
        torch.manual_seed(2)
        a = torch.randn(64, 64, device=device) * 0.2
        x = torch.randn(512, 64, device=device)
        y = x @ a.T
        final = {}
        for skip in (True, False):
            m = self._mk(device, skip=skip)
            opt = torch.optim.Adam(m.parameters(), lr=3e-3)
            for _ in range(120):
                opt.zero_grad()
                o = m(x, y)
                o.loss.backward()
                opt.step()
            final[skip] = float(o.loss.detach())
        assert final[True] < final[False]
