"""Cross-layer crosscoder (SURVEY.md row N4's sibling; reference src/whisper_sae/sae/crosscoder.py,
tests/test_crosscoder.py).

CPU: the oracle's restatement against golden set G15 (seeded init, forward, every gradient, clock, decoder helpers)
produced by the real reference; the host surface of the drop-in module.  GPU: ``TopKCrossLayerCrosscoder`` on the
TopK-SAE kernels against G15 and the oracle, plus the reference's behavioural tests under their original names."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O
from oracle import synth


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def digest(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).reshape(-1)
    with np.errstate(over="ignore"):
        w = (np.arange(u.size, dtype=np.uint64) + np.uint64(1)) * u
        return np.array([u.sum(dtype=np.uint64), w.sum(dtype=np.uint64)], dtype=np.uint64)


@pytest.fixture(scope="module")
def g15(golden_dir):
    return np.load(golden_dir / "g15_crosscoder.npz")


def dims(g):
    return tuple(int(v) for v in g["dims"]), [int(v) for v in g["layers"]]


class TestOracleAgainstReference:
    def test_forward_and_gradients(self, g15):
        (d, L, S, K, B), layers = dims(g15)
        W = (g15["W_enc"], g15["b_enc"], g15["init.W_dec"], g15["b_dec"])
        acts = list(g15["acts"])
        f = O.crosscoder_forward(*W, K, acts)
        assert np.array_equal(np.sort(f["idx"], axis=1).astype(np.int16), g15["idx"])
        assert rel(np.stack(f["recon"]), g15["recon"]) < 1e-5
        assert rel(f["per_layer_loss"], g15["per_layer_loss"]) < 1e-5
        assert abs(float(f["loss"]) - float(g15["loss"])) / float(g15["loss"]) < 1e-5
        assert float(f["l0"]) == float(g15["l0"])
        b = O.crosscoder_backward(*W, acts, f)
        for n in ("W_enc", "b_enc", "W_dec", "b_dec"):
            assert b[n].shape == g15[f"d{n}"].shape
            assert rel(b[n], g15[f"d{n}"]) < 2e-5, n

    def test_is_a_topk_sae_on_the_concatenated_layers(self, g15):
        """The identity the device path rests on: one encoder over [acts_0 | acts_1 | ..] with the MSE divided by
        B * d_model instead of B * L * d_model."""
        (d, L, S, K, B), layers = dims(g15)
        W_e = g15["W_enc"].transpose(2, 0, 1).reshape(S, L * d)
        W_d = g15["init.W_dec"].reshape(S, L * d).T
        x = np.concatenate(list(g15["acts"]), axis=1)
        f = O.transcoder_forward(W_e, g15["b_enc"], W_d, g15["b_dec"].reshape(-1), K, x, x)
        assert abs(float(f["loss"]) * L - float(g15["loss"])) / float(g15["loss"]) < 1e-5
        assert rel(f["predicted"], np.concatenate(list(g15["recon"]), axis=1)) < 1e-5
        b = O.transcoder_backward(W_e, g15["b_enc"], W_d, g15["b_dec"].reshape(-1), x, x, f)
        assert rel(b["W_e"].reshape(S, L, d).transpose(1, 2, 0) * L, g15["dW_enc"]) < 2e-5
        assert rel(b["W_d"].T.reshape(S, L, d) * L, g15["dW_dec"]) < 2e-5


class TestOracleReluAgainstReference:
    def test_forward_and_gradients(self, golden_dir):
        g = np.load(golden_dir / "g16_crosscoder_relu.npz")
        lam = float(g["lam"])
        W = (g["W_enc"], g["b_enc"], g["W_dec"], g["b_dec"])
        acts = list(g["acts"])
        f = O.crosscoder_relu_forward(*W, acts, lam)
        assert rel(np.stack(f["recon"]), g["recon"]) < 1e-5
        assert rel(f["hidden"].sum(axis=1), g["hidden_rowsum"]) < 1e-5
        for key in ("loss", "reconstruction_loss", "sparsity_loss"):
            assert abs(float(f[key]) - float(g[key])) / float(g[key]) < 1e-5, key
        assert float(f["l0"]) == float(g["l0"])
        b = O.crosscoder_relu_backward(*W, acts, f, lam)
        for n in ("W_enc", "b_enc", "W_dec", "b_dec"):
            assert rel(b[n], g[f"d{n}"]) < 2e-5, n


class TestHostSurface:
    def test_seeded_initialisation_equals_the_reference(self, g15):
        from whisper_sae.sae.crosscoder import TopKCrossLayerCrosscoder
        (d, L, S, K, B), layers = dims(g15)
        torch.manual_seed(42)
        m = TopKCrossLayerCrosscoder(d_model=d, n_layers=L, d_sae=S, k=K, layer_indices=layers)
        assert np.array_equal(m.W_dec.detach().numpy(), g15["init.W_dec"])
        assert np.array_equal(digest(m.W_enc.detach().numpy()), g15["init.W_enc_digest"])
        for l in range(L):   # reference test_weight_shapes / _init_weights: encoder = decoder block transposed
            assert torch.equal(m.W_enc[l], m.W_dec[:, l, :].T)
        assert torch.all(m.b_enc == 0) and torch.all(m.b_dec == 0)

    def test_names_shapes_and_factory(self):
        from whisper_sae.sae.crosscoder import (CrosscoderOutput, CrossLayerCrosscoder, TopKCrossLayerCrosscoder,
                                                create_crosscoder)
        m = CrossLayerCrosscoder(d_model=64, n_layers=4, d_sae=256)
        assert (m.d_model, m.n_layers, m.d_sae, m.layer_indices) == (64, 4, 256, [0, 1, 2, 3])
        assert list(m.state_dict().keys()) == ["W_enc", "b_enc", "W_dec", "b_dec", "feature_last_activated", "step_count"]
        assert m.W_enc.shape == (4, 64, 256) and m.b_enc.shape == (256,)
        assert m.W_dec.shape == (256, 4, 64) and m.b_dec.shape == (4, 64)
        assert CrossLayerCrosscoder(64, 2, 128, layer_indices=[1, 3]).layer_indices == [1, 3]
        norms = m.get_decoder_norms()
        assert norms.shape == (256,) and torch.allclose(norms, torch.full_like(norms, 0.1), atol=1e-6)
        assert m.get_feature_layer_norms().shape == (256, 4) and m.get_cross_layer_features().dtype == torch.bool
        t = create_crosscoder(d_model=64, n_layers=4, d_sae=256, k=16, use_topk=True)
        assert isinstance(t, TopKCrossLayerCrosscoder) and t.k == 16 and t.sparsity_weight == 0.0
        r = create_crosscoder(d_model=64, n_layers=4, d_sae=256, use_topk=False)
        assert isinstance(r, CrossLayerCrosscoder) and not isinstance(r, TopKCrossLayerCrosscoder)
        assert create_crosscoder(64, 4, 256, k=16, dead_feature_threshold=500).dead_feature_threshold == 500
        assert CrosscoderOutput._fields == ("reconstructed", "hidden", "loss", "reconstruction_loss", "sparsity_loss",
                                            "l0", "per_layer_loss")

    def test_no_cpu_path_and_no_relu_path(self):
        from whisper_sae import _native as N
        from whisper_sae.sae.crosscoder import CrossLayerCrosscoder, TopKCrossLayerCrosscoder
        acts = {i: torch.randn(4, 32) for i in range(2)}
        with pytest.raises(N.WsaeError):
            TopKCrossLayerCrosscoder(32, 2, 64, k=4)(acts)
        with pytest.raises(N.WsaeError):
            CrossLayerCrosscoder(32, 2, 64)(acts)
        with pytest.raises(N.WsaeError):
            TopKCrossLayerCrosscoder(768, 4, 64, k=4)._check_width()
        with pytest.raises(ValueError):
            TopKCrossLayerCrosscoder(32, 2, 64, k=4).encode({5: torch.randn(4, 32)})


def build(g, device, precision="fp32"):
    from whisper_sae.sae.crosscoder import TopKCrossLayerCrosscoder
    (d, L, S, K, B), layers = dims(g)
    m = TopKCrossLayerCrosscoder(d_model=d, n_layers=L, d_sae=S, k=K, layer_indices=layers, dead_feature_threshold=20,
                                 precision=precision)
    sd = m.state_dict()
    for key, src in (("W_enc", "W_enc"), ("b_enc", "b_enc"), ("W_dec", "init.W_dec"), ("b_dec", "b_dec")):
        sd[key] = torch.from_numpy(np.array(g[src]))
    m.load_state_dict(sd)
    acts = {li: torch.from_numpy(np.array(g["acts"][i])).to(device) for i, li in enumerate(layers)}
    return m.to(device), acts, layers


@pytest.mark.gpu
class TestDeviceAgainstReference:
    def test_forward_gradients_clock_fp32(self, g15, device):
        (d, L, S, K, B), _ = dims(g15)
        m, acts, layers = build(g15, device)
        m.train()
        o = m(acts)
        o.loss.backward()
        assert list(o.reconstructed.keys()) == layers and o.hidden.shape == (B, S)
        assert rel(torch.stack([o.reconstructed[li] for li in layers]).cpu().numpy(), g15["recon"]) < 1e-5
        assert abs(float(o.loss.detach()) - float(g15["loss"])) / float(g15["loss"]) < 1e-5
        assert rel([float(o.per_layer_loss[li]) for li in layers], g15["per_layer_loss"]) < 1e-5
        assert float(o.l0) == float(g15["l0"]) and float(o.sparsity_loss) == 0.0
        h = o.hidden.cpu().numpy()
        assert np.array_equal((h > 0).sum(axis=1).astype(np.int16), g15["nnz"])
        assert np.array_equal(np.sort(np.argsort(-h, axis=1, kind="stable")[:, :K], axis=1).astype(np.int16), g15["idx"])
        for n, p in (("W_enc", m.W_enc), ("b_enc", m.b_enc), ("W_dec", m.W_dec), ("b_dec", m.b_dec)):
            assert p.grad.shape == g15[f"d{n}"].shape
            assert rel(p.grad.cpu().numpy(), g15[f"d{n}"]) < 2e-5, n
        assert int(m.step_count.item()) == int(g15["step_count"])
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), g15["last_activated"])

    def test_helpers_subset_encode_decode_normalize(self, g15, device):
        (d, L, S, K, B), _ = dims(g15)
        m, acts, layers = build(g15, device)
        m.eval()
        assert rel(m.get_decoder_norms().detach().cpu().numpy(), g15["decoder_norms"]) < 1e-6
        assert rel(m.get_feature_layer_norms().detach().cpu().numpy(), g15["layer_norms"]) < 1e-6
        assert np.array_equal(m.get_cross_layer_features(0.5).cpu().numpy(), g15["cross_layer"])
        sub = m.encode({2: acts[2]}).cpu().numpy()
        assert np.array_equal(np.sort(np.argsort(-sub, axis=1, kind="stable")[:, :K], axis=1).astype(np.int16),
                              g15["subset.nnz_idx"])
        assert rel(sub.sum(axis=1), g15["subset.hidden_sum"]) < 1e-5
        full = m.encode(acts)
        dec = m.decode(full)
        assert list(dec.keys()) == layers
        assert rel(torch.stack([dec[li] for li in layers]).cpu().numpy(), g15["recon"]) < 1e-5
        assert int(m.step_count.item()) == 0   # eval mode: the clock stands still
        m.normalize_decoder_weights()
        assert rel(m.W_dec.detach().cpu().numpy(), g15["normalized.W_dec"]) < 1e-6

    def test_padded_width_and_missing_layer(self, device):
        """n_layers * d_model = 72 runs on a 96-column engine; the padded columns never leak into results."""
        from whisper_sae.sae.crosscoder import TopKCrossLayerCrosscoder
        torch.manual_seed(3)
        d, L, S, K, B = 24, 3, 128, 8, 50
        m = TopKCrossLayerCrosscoder(d, L, S, k=K, precision="fp32")
        W = [p.detach().numpy().copy() for p in (m.W_enc, m.b_enc, m.W_dec, m.b_dec)]
        a = [synth.activations(B, d, seed=31, stream=i, bf16=False) for i in range(L)]
        m.to(device).train()
        o = m({i: torch.from_numpy(a[i]).to(device) for i in range(L)})
        o.loss.backward()
        f = O.crosscoder_forward(*W, K, a)
        b = O.crosscoder_backward(*W, a, f)
        assert abs(float(o.loss.detach()) - float(f["loss"])) / float(f["loss"]) < 1e-5
        assert rel(torch.stack([o.reconstructed[i] for i in range(L)]).cpu().numpy(), np.stack(f["recon"])) < 1e-5
        for n, p in (("W_enc", m.W_enc), ("b_enc", m.b_enc), ("W_dec", m.W_dec), ("b_dec", m.b_dec)):
            assert rel(p.grad.cpu().numpy(), b[n]) < 2e-5, n
        with pytest.raises(KeyError):   # forward needs every layer (the reference indexes the dict, crosscoder.py:355)
            m({0: torch.from_numpy(a[0]).to(device)})

    def test_whisper_tiny_dimensions_bf16_against_the_oracle(self, device):
        """Reference test_whisper_tiny_dimensions at a real batch: d_model 384 x 4 layers, d_sae 3072, k 32, in the
        bf16 mode; the oracle rounds the same operands to bf16 and selects the device's features."""
        from whisper_sae.sae.crosscoder import TopKCrossLayerCrosscoder
        torch.manual_seed(5)
        d, L, S, K, B = 384, 4, 3072, 32, 1024
        m = TopKCrossLayerCrosscoder(d, L, S, k=K, precision="bf16")
        W = [p.detach().numpy().copy() for p in (m.W_enc, m.b_enc, m.W_dec, m.b_dec)]
        a = [synth.activations(B, d, seed=37, stream=i, bf16=True) for i in range(L)]
        m.to(device).train()
        o = m({i: torch.from_numpy(a[i]).to(device) for i in range(L)})
        o.loss.backward()
        assert o.hidden.shape == (B, S) and all(o.reconstructed[i].shape == (B, d) for i in range(L))
        assert float(o.l0) <= K
        _, idx = m._last_code
        f = O.crosscoder_forward(*W, K, a, mode="amp", select=idx.cpu().numpy())
        assert O.check_selection(f["pre"], idx.cpu().numpy().astype(np.int64), K, rtol=1e-4).all()
        assert abs(float(o.loss.detach()) - float(f["loss"])) / float(f["loss"]) < 2e-3
        assert rel(torch.stack([o.reconstructed[i] for i in range(L)]).cpu().numpy(), np.stack(f["recon"])) < 2e-2
        b = O.crosscoder_backward(*W, a, f)
        for n, p in (("W_enc", m.W_enc), ("b_enc", m.b_enc), ("W_dec", m.W_dec), ("b_dec", m.b_dec)):
            assert rel(p.grad.cpu().numpy(), b[n]) < 3e-2, n


@pytest.mark.gpu
class TestReluDeviceAgainstReference:
    """``CrossLayerCrosscoder(activation="relu")`` on the ReLU-SAE kernels with the decoder norms as L1 weights."""

    def build(self, g, device, precision="fp32"):
        from whisper_sae.sae.crosscoder import CrossLayerCrosscoder
        d, L, S, B = (int(v) for v in g["dims"])
        layers = [int(v) for v in g["layers"]]
        m = CrossLayerCrosscoder(d_model=d, n_layers=L, d_sae=S, layer_indices=layers, sparsity_weight=float(g["lam"]),
                                 dead_feature_threshold=20, precision=precision)
        sd = m.state_dict()
        for key in ("W_enc", "b_enc", "W_dec", "b_dec"):
            sd[key] = torch.from_numpy(np.array(g[key]))
        m.load_state_dict(sd)
        acts = {li: torch.from_numpy(np.array(g["acts"][i])).to(device) for i, li in enumerate(layers)}
        return m.to(device), acts, layers, (d, L, S, B)

    def test_forward_gradients_clock_fp32(self, golden_dir, device):
        g = np.load(golden_dir / "g16_crosscoder_relu.npz")
        m, acts, layers, (d, L, S, B) = self.build(g, device)
        m.train()
        o = m(acts)
        o.loss.backward()
        assert rel(torch.stack([o.reconstructed[li] for li in layers]).cpu().numpy(), g["recon"]) < 1e-5
        assert rel(o.hidden.sum(dim=1).cpu().numpy(), g["hidden_rowsum"]) < 1e-5
        for key, val in (("loss", o.loss), ("reconstruction_loss", o.reconstruction_loss), ("sparsity_loss", o.sparsity_loss)):
            assert abs(float(val.detach()) - float(g[key])) / float(g[key]) < 1e-5, key
        assert rel([float(o.per_layer_loss[li]) for li in layers], g["per_layer_loss"]) < 1e-5
        assert float(o.l0) == float(g["l0"])
        for n, p in (("W_enc", m.W_enc), ("b_enc", m.b_enc), ("W_dec", m.W_dec), ("b_dec", m.b_dec)):
            assert p.grad.shape == g[f"d{n}"].shape
            assert rel(p.grad.cpu().numpy(), g[f"d{n}"]) < 2e-5, n
        assert int(m.step_count.item()) == int(g["step_count"])
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), g["last_activated"])
        # the module leaves the ctx as the ReLU SAE expects it: a second forward is the same forward
        o2 = m(acts)
        assert float(o2.loss.detach()) == float(o.loss.detach())

    def test_sparsity_loss_uses_decoder_norms(self, device):   # reference test of the same name
        from whisper_sae.sae.crosscoder import CrossLayerCrosscoder
        torch.manual_seed(1)
        m = CrossLayerCrosscoder(d_model=64, n_layers=4, d_sae=256, sparsity_weight=0.1, precision="fp32").to(device)
        acts = {i: torch.randn(8, 64, device=device) for i in range(4)}
        out = m(acts)
        expected = torch.mean(out.hidden.abs() @ m.get_decoder_norms().detach())
        assert torch.isclose(out.sparsity_loss, expected, rtol=1e-5)
        assert torch.isclose(out.loss, out.reconstruction_loss + 0.1 * out.sparsity_loss, rtol=1e-5)
        hidden = m.encode({0: acts[0]})   # reference test_encode_combines_layers: a subset of the layers
        assert hidden.shape == (8, 256) and torch.all(hidden >= 0)

    def test_bf16_mode_against_the_oracle(self, device):
        from whisper_sae.sae.crosscoder import CrossLayerCrosscoder
        torch.manual_seed(2)
        d, L, S, B, lam = 64, 4, 512, 256, 0.02
        m = CrossLayerCrosscoder(d_model=d, n_layers=L, d_sae=S, sparsity_weight=lam, precision="bf16")
        W = [p.detach().numpy().copy() for p in (m.W_enc, m.b_enc, m.W_dec, m.b_dec)]
        a = [synth.activations(B, d, seed=41, stream=i, bf16=True) for i in range(L)]
        m.to(device).train()
        o = m({i: torch.from_numpy(a[i]).to(device) for i in range(L)})
        o.loss.backward()
        f = O.crosscoder_relu_forward(*W, a, lam)
        b = O.crosscoder_relu_backward(*W, a, f, lam)
        assert abs(float(o.loss.detach()) - float(f["loss"])) / float(f["loss"]) < 5e-3
        assert abs(float(o.sparsity_loss) - float(f["sparsity_loss"])) / float(f["sparsity_loss"]) < 5e-3
        # (the oracle here is plain fp32 arithmetic, not a mirror of the bf16 roundings: direction and size, not digits)
        for n, p in (("W_enc", m.W_enc), ("b_enc", m.b_enc), ("W_dec", m.W_dec), ("b_dec", m.b_dec)):
            got, ref = p.grad.cpu().numpy().astype(np.float64).ravel(), b[n].astype(np.float64).ravel()
            cos = float(got @ ref / np.sqrt((got @ got) * (ref @ ref)))
            assert cos > 0.999 and abs(np.linalg.norm(got) / np.linalg.norm(ref) - 1) < 2e-2, (n, cos)

    def test_loss_decreases(self, device):   # reference TestCrosscoderTraining.test_loss_decreases, ReLU variant
        from whisper_sae.sae.crosscoder import CrossLayerCrosscoder
        torch.manual_seed(0)
        m = CrossLayerCrosscoder(d_model=32, n_layers=2, d_sae=128, sparsity_weight=0.01).to(device)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        base = torch.randn(64, 32, device=device)
        acts = {0: base + 0.1 * torch.randn(64, 32, device=device), 1: base + 0.1 * torch.randn(64, 32, device=device)}
        losses = []
        for _ in range(100):
            opt.zero_grad()
            out = m(acts)
            out.loss.backward()
            opt.step()
            m.normalize_decoder_weights()
            losses.append(out.loss.item())
        assert losses[-1] < losses[0] * 0.8


@pytest.mark.gpu
class TestReferenceBehaviour:
    """The reference's tests/test_crosscoder.py cases that exercise the compute path, same names."""

    def make(self, device, **kw):
        from whisper_sae.sae.crosscoder import TopKCrossLayerCrosscoder
        args = dict(d_model=64, n_layers=4, d_sae=256, k=16)
        args.update(kw)
        return TopKCrossLayerCrosscoder(**args).to(device)

    def test_topk_sparsity(self, device):
        m = self.make(device)
        hidden = m.encode({i: torch.randn(8, 64, device=device) for i in range(4)})
        assert torch.all((hidden > 0).sum(dim=-1) <= 16)

    def test_l0_equals_k(self, device):
        m = self.make(device)
        out = m({i: torch.randn(8, 64, device=device).abs() + 0.1 for i in range(4)})
        assert out.l0.item() <= 16

    def test_no_sparsity_loss(self, device):
        m = self.make(device)
        out = m({i: torch.randn(8, 64, device=device) for i in range(4)})
        assert out.sparsity_loss.item() == 0.0
        assert torch.isclose(out.loss, out.reconstruction_loss)

    def test_reconstruction_loss_is_sum_of_layers(self, device):
        m = self.make(device)
        out = m({i: torch.randn(8, 64, device=device) for i in range(4)})
        assert torch.isclose(out.reconstruction_loss, sum(out.per_layer_loss.values()), rtol=1e-5)

    def test_dead_feature_tracking(self, device):
        m = self.make(device)
        m.train()
        for _ in range(5):
            m({i: torch.randn(8, 64, device=device) for i in range(4)})
        assert m.step_count.item() == 5
        dead = m.get_dead_features()
        assert dead.shape == (256,) and dead.dtype == torch.bool
        assert 0.0 <= m.get_dead_feature_ratio() <= 1.0

    def test_gradients_flow(self, device):
        m = self.make(device)
        m({i: torch.randn(8, 64, device=device) for i in range(4)}).loss.backward()
        for p in (m.W_enc, m.W_dec, m.b_enc, m.b_dec):
            assert p.grad is not None and torch.isfinite(p.grad).all()

    def test_loss_decreases(self, device):
        torch.manual_seed(0)
        m = self.make(device, d_model=32, n_layers=2, d_sae=128, k=8)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        base = torch.randn(64, 32, device=device)
        acts = {0: base + 0.1 * torch.randn(64, 32, device=device), 1: base + 0.1 * torch.randn(64, 32, device=device)}
        losses = []
        for _ in range(100):
            opt.zero_grad()
            out = m(acts)
            out.loss.backward()
            opt.step()
            m.normalize_decoder_weights()
            losses.append(out.loss.item())
        assert losses[-1] < losses[0] * 0.8

    def test_subset_of_layers(self, device):
        m = self.make(device, d_model=384, n_layers=2, d_sae=512, k=32, layer_indices=[1, 3])
        out = m({1: torch.randn(16, 384, device=device), 3: torch.randn(16, 384, device=device)})
        assert set(out.reconstructed.keys()) == {1, 3} and set(out.per_layer_loss.keys()) == {1, 3}
