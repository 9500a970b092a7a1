"""Activation producer (reference src/whisper_sae/sae/hooks.py, tests/test_hooks.py).

The reference's tests load ``openai/whisper-tiny`` from the hub; there is no network here, so the model is a seeded
random-init Whisper of the same architecture family (``make_golden.py:tiny_whisper``) and golden set G14 holds what
the reference's ``extract_features_batch`` returns for it.  CPU: the drop-in module against G14 and the reference's
own behavioural tests (hook registration / removal, shapes, LayerNorm on/off, accumulation).  GPU: the hooked
activations go through ``wsae_ring_push_layernorm`` into the on-device ring and equal the reference's
layer-normed, flattened activations; the kernel alone against ``torch.nn.LayerNorm`` on ragged widths."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from whisper_sae.sae.hooks import ActivationCache, WhisperActivationExtractor, extract_features_batch, flatten_activations

ROOT = __import__("pathlib").Path(__file__).resolve().parents[1]


def tiny_whisper(seed: int = 0):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(vocab_size=200, num_mel_bins=80, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
                        decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128, d_model=64,
                        max_source_positions=50, max_target_positions=16, decoder_start_token_id=1, pad_token_id=0,
                        bos_token_id=1, eos_token_id=2)
    torch.manual_seed(seed)
    return WhisperForConditionalGeneration(cfg).eval()


@pytest.fixture(scope="module")
def g14(golden_dir):
    return np.load(golden_dir / "g14_hooks.npz")


@pytest.fixture(scope="module")
def model():
    return tiny_whisper(0)


def versions_match(g):
    import transformers
    return str(g["transformers_version"]) == transformers.__version__ and str(g["torch_version"]) == torch.__version__


class TestAgainstReference:
    @pytest.mark.parametrize("tag,ln", [("ln", True), ("raw", False)])
    def test_extract_features_batch_matches_g14(self, g14, model, tag, ln):
        if not versions_match(g14):
            pytest.skip("G14 was generated with another transformers / torch build")
        r = extract_features_batch(model, torch.from_numpy(g14["mel"]), [0, 1], [0, 1], ln, "cpu")
        for comp in ("encoder", "decoder"):
            for layer in (0, 1):
                want = g14[f"{tag}.{comp}.{layer}"]
                got = r[comp][layer].numpy()
                assert got.shape == want.shape
                assert np.array_equal(got, want), (comp, layer)  # same torch ops, same machine arithmetic
        assert np.array_equal(flatten_activations(r["encoder"][1], "encoder").numpy(), g14[f"{tag}.flat"])


class TestReferenceBehaviour:
    """tests/test_hooks.py of the reference, on the offline model."""

    def test_cache(self):
        c = ActivationCache()
        assert c.encoder == {} and c.decoder == {} and c.get_encoder_activations(0) is None
        c.encoder[0] = [torch.ones(2, 3, 4), torch.zeros(1, 3, 4)]
        c.decoder[1] = [torch.ones(2, 1, 4)]
        assert c.get_encoder_activations(0).shape == (3, 3, 4) and c.get_decoder_activations(1).shape == (2, 1, 4)
        c.clear()
        assert c.encoder == {} and c.decoder == {}

    def test_hook_registration_and_removal(self, model):
        ex = WhisperActivationExtractor(model, encoder_layers=[0, 1], decoder_layers=[0])
        assert ex.apply_layer_norm and len(ex._hooks) == 0
        ex.register_hooks()
        assert len(ex._hooks) == 3
        ex.remove_hooks()
        assert len(ex._hooks) == 0
        with ex:
            assert len(ex._hooks) == 3
        assert len(ex._hooks) == 0

    def test_shapes_and_accumulation(self, model):
        x = torch.randn(2, 80, 100)
        ex = WhisperActivationExtractor(model, encoder_layers=[0, 1], decoder_layers=[])
        with torch.no_grad(), ex:
            model.model.encoder(x)
            model.model.encoder(x)
        a = ex.cache.get_encoder_activations(1)
        assert a.shape == (4, 50, 64)  # two batches of 2, 100 mel frames -> 50 positions
        ex.clear_cache()
        assert ex.cache.get_encoder_activations(1) is None

    def test_layer_norm_is_the_models_final_norm(self, model):
        x = torch.randn(1, 80, 100)
        raw = extract_features_batch(model, x, [1], [], apply_layer_norm=False)["encoder"][1]
        ln = extract_features_batch(model, x, [1], [], apply_layer_norm=True)["encoder"][1]
        assert torch.allclose(ln, model.model.encoder.layer_norm(raw), atol=1e-6)
        # the last layer's normed output is the encoder's output
        with torch.no_grad():
            assert torch.allclose(ln, model.model.encoder(x).last_hidden_state, atol=1e-5)

    def test_empty_layer_lists_and_flatten(self, model):
        r = extract_features_batch(model, torch.randn(1, 80, 100), [], [])
        assert r == {"encoder": {}, "decoder": {}}
        t = torch.arange(24.0).reshape(2, 3, 4)
        f = flatten_activations(t, "decoder")
        assert f.shape == (6, 4) and torch.equal(f[4], t[1, 1])


@pytest.fixture(scope="module")
def g17(golden_dir):
    return np.load(golden_dir / "g17_extraction.npz")


class TestExtractionDriver:
    """``extract_and_cache_features`` (reference data/feature_cache.py:200-306) against golden set G17: what the
    reference's driver wrote for the seeded tiny Whisper - cache tensors and every field of the metadata sidecars."""

    def _batches(self, g17):
        mel = g17["mel"]
        return [torch.from_numpy(mel[i:i + 2]) for i in range(0, mel.shape[0], 2)]

    def test_cache_files_match_g17(self, g17, tmp_path, capsys):
        import json
        from whisper_sae.config import DataConfig, WhisperConfig
        from whisper_sae.data.feature_cache import FeatureCache, extract_and_cache_features
        if not versions_match(g17):
            pytest.skip("G17 was generated with another transformers / torch build")
        fc = FeatureCache(tmp_path, WhisperConfig(), DataConfig(cache_dir="cache"))
        r = extract_and_cache_features(tiny_whisper(0), None, self._batches(g17), fc, [0, 1], [], device="cpu", max_samples=5,
                                       show_progress=False)
        assert r["num_samples"] == 6  # whole batches while fewer than max_samples are in: 2 + 2 + 2 (feature_cache.py:256-258)
        assert sorted(p.name for p in tmp_path.iterdir()) == list(g17["files"])
        for layer in (0, 1):
            feats, meta = fc.load("encoder", layer)
            assert np.array_equal(feats.numpy(), g17[f"encoder.{layer}"])  # same torch ops on the same machine arithmetic
            want = json.loads(str(g17[f"meta.encoder.{layer}"]))
            got = json.loads(meta.to_json())
            got["created_at"] = ""
            assert got == want
            assert r["tokens"][("encoder", layer)] == want["num_tokens"]
        assert "Cached encoder layer 0" in capsys.readouterr().out

    def test_tuple_batches_and_argument_checks(self, g17, tmp_path):
        from whisper_sae.config import DataConfig, WhisperConfig
        from whisper_sae.data.feature_cache import FeatureCache, extract_and_cache_features
        fc = FeatureCache(tmp_path, WhisperConfig(), DataConfig(cache_dir="cache"))
        batches = [(b, "ignored") for b in self._batches(g17)[:1]]  # (features, ...) tuples as a DataLoader yields them
        r = extract_and_cache_features(tiny_whisper(0), None, batches, fc, [1], [], device="cpu", show_progress=False)
        assert r["num_samples"] == 2 and fc.has_cache("encoder", 1) and not fc.has_cache("encoder", 0)
        with pytest.raises(ValueError):  # a ring for a layer that is not extracted
            extract_and_cache_features(tiny_whisper(0), None, batches, fc, [1], [], rings={("encoder", 0): object()}, show_progress=False)
        with pytest.raises(ValueError):  # no cache and no ring for a layer
            extract_and_cache_features(tiny_whisper(0), None, batches, None, [1], [], show_progress=False)

    def test_decoder_layers_fail_like_the_reference_on_this_transformers(self, g17, tmp_path):
        # G17 records that the reference's driver raises ValueError for decoder layers with this transformers build (its hook
        # takes output[0] of a bare tensor, hooks.py:99-101, and flatten_activations then meets a 2-D tensor): same here
        from whisper_sae.config import DataConfig, WhisperConfig
        from whisper_sae.data.feature_cache import FeatureCache, extract_and_cache_features
        if not versions_match(g17) or str(g17["decoder_raises"]) != "ValueError":
            pytest.skip("recorded for another transformers build")
        fc = FeatureCache(tmp_path, WhisperConfig(), DataConfig(cache_dir="cache"))
        with pytest.raises(ValueError):
            extract_and_cache_features(tiny_whisper(0), None, self._batches(g17)[:1], fc, [], [1], device="cpu", show_progress=False)


@pytest.mark.gpu
class TestIntoTheRing:
    def test_layernorm_push_kernel(self, device):
        from whisper_sae.data.feature_cache import ActivationRing
        torch.manual_seed(1)
        for dim, rows, dt in ((384, 1000, torch.float32), (64, 37, torch.float32), (1280, 130, torch.bfloat16), (96, 5, torch.bfloat16)):
            ring = ActivationRing(2048, dim, device=device, dtype=torch.float32)
            norm = torch.nn.LayerNorm(dim).to(device)
            with torch.no_grad():
                norm.weight.normal_(1.0, 0.2)
                norm.bias.normal_(0.0, 0.2)
            h = (torch.randn(rows, dim, device=device) * 3 + 1).to(dt)
            ring.push_layernorm(h, norm.weight, norm.bias, norm.eps)
            torch.cuda.synchronize()
            want = norm(h.float())
            assert len(ring) == rows
            assert torch.allclose(ring.data[:rows], want, rtol=1e-5, atol=2e-5), (dim, rows)
            ring.close()

    def test_bf16_ring_wraps_around(self, device):
        from whisper_sae.data.feature_cache import ActivationRing
        ring = ActivationRing(100, 64, device=device, dtype=torch.bfloat16)
        norm = torch.nn.LayerNorm(64).to(device)
        a, b = torch.randn(70, 64, device=device), torch.randn(60, 64, device=device)
        ring.push_layernorm(a, norm.weight, norm.bias, norm.eps)
        ring.push_layernorm(b, norm.weight, norm.bias, norm.eps)
        torch.cuda.synchronize()
        assert len(ring) == 100
        wb = norm(b).to(torch.bfloat16)
        assert torch.equal(ring.data[70:100], wb[:30]) and torch.equal(ring.data[0:30], wb[30:])  # newest rows overwrite the oldest

    def test_hooked_whisper_feeds_the_ring_the_trainer_samples(self, g14, device):
        from whisper_sae.data.feature_cache import ActivationRing
        model = tiny_whisper(0).to(device)
        mel = torch.from_numpy(g14["mel"]).to(device)
        ring = ActivationRing(4096, 64, device=device, dtype=torch.float32)
        r = extract_features_batch(model, mel, [0, 1], [1], True, device, rings={("encoder", 1): ring}, keep_on_device=True)
        torch.cuda.synchronize()
        assert 1 not in r["encoder"] and 0 in r["encoder"] and r["encoder"][0].is_cuda  # layer 1 went to the ring, layer 0 stayed on the GPU
        r_host = extract_features_batch(model, mel, [0], [], True, device)  # default: the reference's .cpu() (hooks.py:92)
        assert not r_host["encoder"][0].is_cuda and torch.equal(r_host["encoder"][0], r["encoder"][0].cpu())
        ring.clear() if hasattr(ring, "clear") else None
        ring2 = ActivationRing(4096, 64, device=device, dtype=torch.float32)
        extract_features_batch(model, mel, [1], [], True, device, rings={("encoder", 1): ring2})
        torch.cuda.synchronize()
        ring = ring2
        want = torch.from_numpy(g14["ln.flat"]).to(device)  # the reference's layer-normed, flattened activations (CPU)
        assert len(ring) == want.shape[0]
        assert torch.allclose(ring.data[:len(ring)], want, rtol=2e-4, atol=2e-4)  # GPU vs CPU attention arithmetic
        # ... and an SAE trains straight from it
        from whisper_sae.config import TrainingConfig
        from whisper_sae.data import RingLoader
        from whisper_sae.sae.model import TopKSAE
        from whisper_sae.sae.training import SAETrainer
        sae = TopKSAE(64, 256, k=8)
        tr = SAETrainer(sae, TrainingConfig(batch_size=64, use_amp=False, num_workers=0), device=device)
        m = tr.train_step(next(iter(RingLoader(ring, 64, shuffle=True, seed=1))))
        assert np.isfinite(m.loss) and m.l0 == 8

    def test_producer_ring_trainer_in_one_pass(self, g17, device, tmp_path):
        """hooks -> ring -> 50 train steps: the driver pushes the layer-normed encoder activations of
        the seeded tiny Whisper straight into ActivationRing objects, nothing is written, and an SAE trained on the ring's
        rows lowers its loss."""
        from whisper_sae.config import TrainingConfig
        from whisper_sae.data import ActivationRing, RingLoader, extract_and_cache_features
        from whisper_sae.sae.model import TopKSAE
        from whisper_sae.sae.training import SAETrainer
        model = tiny_whisper(0).to(device)
        mel = torch.from_numpy(g17["mel"])
        batches = [mel[i:i + 2] for i in range(0, 8, 2)]
        rings = {("encoder", 0): ActivationRing(400, 64, device=device, dtype=torch.float32),
                 ("encoder", 1): ActivationRing(400, 64, device=device, dtype=torch.float32)}
        r = extract_and_cache_features(model, None, batches, None, [0, 1], [], device=device, rings=rings, show_progress=False)
        torch.cuda.synchronize()
        assert r["num_samples"] == 8 and r["tokens"] == {("encoder", 0): 400, ("encoder", 1): 400}
        assert not any(tmp_path.iterdir())
        want = torch.from_numpy(g17["encoder.1"]).to(device)  # the reference's first 6 clips (CPU arithmetic)
        assert torch.allclose(rings[("encoder", 1)].data[:300], want, rtol=2e-4, atol=2e-4)
        sae = TopKSAE(64, 512, k=8)
        tr = SAETrainer(sae, TrainingConfig(batch_size=100, learning_rate=3e-3, warmup_steps=0, use_amp=False, num_workers=0),
                        device=device, run_dir=tmp_path / "run")
        loader = RingLoader(rings[("encoder", 1)], 100, shuffle=True, seed=3)
        losses = []
        while len(losses) < 50:
            for b in loader:
                losses.append(tr.train_step(b).loss)
        assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])

    def test_cli_extract_only_then_train_and_stream_to_ring(self, device, tmp_path):
        """scripts/train.py: --extract-only writes the reference's cache files from a model OBJECT; a second run trains from
        them; --stream-to-ring extracts and trains in one process without cache files (reference scripts/train.py:283-329)."""
        import importlib.util
        import yaml
        spec = importlib.util.spec_from_file_location("wsae_train_cli", ROOT / "scripts" / "train.py")
        cli = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cli)
        cfg = {"whisper": {"model_name": "local/tiny-random", "hidden_dim": 64, "num_encoder_layers": 2, "num_decoder_layers": 2},
               "sae": {"expansion_factor": 4, "k": 8, "dead_feature_resample": False},
               "training": {"batch_size": 64, "epochs": 2, "use_amp": False, "num_workers": 0, "warmup_steps": 0, "checkpoint_every": 1},
               "data": {"cache_dir": str(tmp_path / "cache"), "max_samples": 6}, "wandb": {"enabled": False},
               "encoder_layers": [1], "decoder_layers": [], "output_dir": str(tmp_path / "out"), "experiment_name": "n2"}
        (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
        mel = torch.randn(6, 80, 100, generator=torch.Generator().manual_seed(5))
        torch.save(mel, tmp_path / "mel.pt")
        base = ["--config", str(tmp_path / "cfg.yaml"), "--no-wandb", "--device", str(device), "--mel", str(tmp_path / "mel.pt")]
        cli.main(base + ["--extract-only"], whisper_model=tiny_whisper(0))
        feats = tmp_path / "cache" / "features" / "tiny-random_encoder_layer1.pt"
        assert feats.exists() and (tmp_path / "cache" / "features" / "tiny-random_encoder_layer1_meta.json").exists()
        assert tuple(torch.load(feats, weights_only=True).shape) == (300, 64)
        assert not (tmp_path / "out").exists()  # extract-only: no training
        cli.main(base)  # cache present: trains from it without a model
        run = tmp_path / "out" / "n2_encoder_layer1"
        assert (run / "final.pt").exists() and (run / "sae_final.pt").exists() and (run / "metrics.json").exists()
        # one process, no files: extraction pushes into the ring the trainer samples
        cfg["data"]["cache_dir"] = str(tmp_path / "cache2")
        cfg["experiment_name"] = "n2ring"
        (tmp_path / "cfg2.yaml").write_text(yaml.safe_dump(cfg))
        cli.main(["--config", str(tmp_path / "cfg2.yaml"), "--no-wandb", "--device", str(device), "--mel", str(tmp_path / "mel.pt"),
                  "--stream-to-ring"], whisper_model=tiny_whisper(0))
        assert (tmp_path / "out" / "n2ring_encoder_layer1" / "final.pt").exists()
        assert not list((tmp_path / "cache2" / "features").glob("*.pt"))
        with pytest.raises(SystemExit):  # --extract-only without a model: this build never downloads one by name
            cli.main(["--config", str(tmp_path / "cfg2.yaml"), "--no-wandb", "--device", str(device), "--extract-only"])
