"""CPU pins that round 1 claimed but did not commit: the torch-CPU restatement that ``bench.py`` times
as ``cpu_baseline`` against the reference's own train steps (G3 / G4), the drop-in modules' seeded initialisation
against the reference's (G10), and the activation-cache interchange with the reference's ``FeatureCache`` (G11, N1)."""

from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import synth
from oracle.torch_step import TorchCPUStep

REFERENCE_SRC = Path("/root/reference/src")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


class TestTorchCPUStep:
    """oracle/torch_step.py is the op sequence the reference issues per step; here it reproduces the reference."""

    def test_g3_one_step(self, golden_dir):
        g1 = np.load(golden_dir / "g1_forward_cfg2.npz")
        g3 = np.load(golden_dir / "g3_train_step_cfg2.npz")
        D, H, K, B = (int(v) for v in g1["dims"])
        w = synth.sae_weights(D, H, seed=42, bf16=True, b_pre_scale=0.1)
        x = torch.from_numpy(synth.activations(B, D, seed=42, stream=1, bf16=True))
        torch.set_num_threads(4)
        step = TorchCPUStep(w, K, lr=float(g3["lr0"]), weight_decay=0.0, max_norm=1.0, dead_feature_threshold=1000)
        out = step.step(x)
        assert out["loss"] == float(g3["loss"])          # same ops, same order: bit-equal on the same torch build
        assert out["l0"] == float(g3["l0"])
        assert out["dead_feature_ratio"] == float(g3["dead_ratio"])
        assert abs(out["grad_norm"] - float(g3["grad_total_norm"])) < 1e-5 * float(g3["grad_total_norm"])  # clip_grad_norm_ sums in fp32
        assert np.array_equal(step.b_e.detach().numpy(), g3["b_e"])
        assert np.array_equal(step.b_d.detach().numpy(), g3["b_d"])
        assert np.array_equal(step.b_pre.detach().numpy(), g3["b_pre"])
        assert np.array_equal(step.W_e.detach().numpy().reshape(-1)[g3["pos_e"]], g3["W_e_samples"])
        assert np.array_equal(step.W_d.detach().numpy().reshape(-1)[g3["pos_d"]], g3["W_d_samples"])
        # the reference's steps 2 and 3 (tuple / list batch forms of the same x) with the scheduler's next rates
        for s in (1, 2):
            o = step.step(x, lr=float(g3["lrs_3steps"][s - 1]))
            assert abs(o["loss"] - g3["losses_3steps"][s]) <= 1e-6 * g3["losses_3steps"][s]

    def test_g4_trajectory(self, golden_dir):
        g = np.load(golden_dir / "g4_trajectory_small.npz")
        D, H, K, B, STEPS = (int(v) for v in g["dims"])
        w = synth.sae_weights(D, H, seed=7, bf16=False, b_pre_scale=0.05)
        xs = synth.activations(B * STEPS, D, seed=7, stream=2, bf16=False).reshape(STEPS, B, D)
        torch.set_num_threads(4)
        step = TorchCPUStep(w, K, lr=1e-3, weight_decay=0.01, max_norm=1.0, dead_feature_threshold=5)
        for s in range(STEPS):
            o = step.step(torch.from_numpy(xs[s]), lr=float(g["lrs"][s]))
            assert abs(o["loss"] - g["losses"][s]) <= 1e-6 * g["losses"][s], s
            assert o["dead_feature_ratio"] == pytest.approx(g["dead"][s], abs=1e-7), s
        assert rel(step.W_e.detach().numpy(), g["W_e"]) < 1e-6
        assert rel(step.W_d.detach().numpy(), g["W_d"]) < 1e-6
        assert rel(step.b_pre.detach().numpy(), g["b_pre"]) < 1e-6
        assert np.array_equal(step.last_activated.numpy(), g["last_activated"])
        assert int(step.step_count) == int(g["step_count"])


def _digest(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).reshape(-1)
    with np.errstate(over="ignore"):
        w = (np.arange(u.size, dtype=np.uint64) + np.uint64(1)) * u
        return np.array([u.sum(dtype=np.uint64), w.sum(dtype=np.uint64)], dtype=np.uint64)


class TestSeededInit:
    """same construction and RNG draw order as the reference (model.py:38-89, :266-286), so
    ``torch.manual_seed(s)`` followed by ``TopKSAE(...)`` / ``ReLUSAE(...)`` yields the reference's parameters."""

    @pytest.mark.parametrize("tag,D,H,K", [("cfg2", 384, 3072, 32), ("small", 64, 256, 8)])
    def test_same_parameters_as_the_reference(self, golden_dir, tag, D, H, K):
        from whisper_sae.sae.model import ReLUSAE, TopKSAE
        g = np.load(golden_dir / "g10_seeded_init.npz")
        if str(g["torch_version"]) != torch.__version__:
            pytest.skip(f"golden drawn with torch {g['torch_version']}, running {torch.__version__}")
        torch.manual_seed(42)
        m = TopKSAE(D, H, k=K)
        for k_, v in m.state_dict().items():
            if v.dtype == torch.float32:
                a = v.detach().numpy()
                assert np.array_equal(a.reshape(-1)[:64], g[f"topk.{tag}.{k_}.head"]), k_
                assert np.array_equal(_digest(a), g[f"topk.{tag}.{k_}.digest"]), k_
        assert np.array_equal(torch.rand(4).numpy(), g[f"topk.{tag}.next_rand"])  # drew exactly as many numbers
        torch.manual_seed(42)
        r = ReLUSAE(D, H, sparsity_weight=0.01)
        for k_, v in r.state_dict().items():
            a = v.detach().numpy()
            assert np.array_equal(_digest(a), g[f"relu.{tag}.{k_}.digest"]), k_
        assert np.array_equal(torch.rand(4).numpy(), g[f"relu.{tag}.next_rand"])


class TestCacheInterchange:
    """``--extract-only`` caches are interchangeable with the reference (feature_cache.py:87-167)."""

    def test_reads_a_cache_the_reference_wrote(self, golden_dir):
        from whisper_sae.config import DataConfig, WhisperConfig
        from whisper_sae.data import FeatureCache
        fc = FeatureCache(golden_dir / "g11_cache", WhisperConfig(), DataConfig())
        assert fc.has_cache("encoder", 0) and not fc.has_cache("encoder", 1)
        feats, meta = fc.load("encoder", 0)
        want = synth.activations(96, 384, seed=77, stream=0, bf16=False)
        assert feats.dtype == torch.float32 and np.array_equal(feats.numpy(), want)
        assert (meta.model_name, meta.component, meta.layer_idx) == ("openai/whisper-tiny", "encoder", 0)
        assert (meta.hidden_dim, meta.num_samples, meta.num_tokens) == (384, 2, 96)
        assert meta.data_config["dataset_name"] == "librispeech_asr"

    def test_sidecar_has_the_reference_fields(self, golden_dir, tmp_path):
        from whisper_sae.config import DataConfig, WhisperConfig
        from whisper_sae.data import FeatureCache
        fc = FeatureCache(tmp_path, WhisperConfig(), DataConfig(cache_dir=Path("cache")))
        fc.save(torch.from_numpy(synth.activations(8, 384, seed=1, stream=0, bf16=False)), "encoder", 0, num_samples=1)
        mine = json.loads(fc._get_metadata_path("encoder", 0).read_text())
        ref = json.loads((golden_dir / "g11_cache" / "whisper-tiny_encoder_layer0_meta.json").read_text())
        assert list(mine.keys()) == list(ref.keys())
        assert mine["data_config"].keys() == ref["data_config"].keys()
        assert fc._get_cache_path("encoder", 0).name == "whisper-tiny_encoder_layer0.pt"

    @pytest.mark.skipif(not REFERENCE_SRC.exists(), reason="the reference only exists in the build container")
    def test_the_reference_reads_a_cache_this_build_wrote(self, tmp_path):
        from whisper_sae.config import DataConfig, WhisperConfig
        from whisper_sae.data import FeatureCache
        feats = torch.from_numpy(synth.activations(40, 384, seed=9, stream=0, bf16=False))
        FeatureCache(tmp_path, WhisperConfig(), DataConfig(cache_dir=Path("cache"))).save(feats, "decoder", 3, num_samples=5)
        code = (
            "import sys, json, torch\n"
            "from pathlib import Path\n"
            "from whisper_sae.config import DataConfig, WhisperConfig\n"
            "from whisper_sae.data.feature_cache import FeatureCache\n"
            "fc = FeatureCache(Path(sys.argv[1]), WhisperConfig(), DataConfig())\n"
            "assert fc.has_cache('decoder', 3)\n"
            "f, m = fc.load('decoder', 3)\n"
            "print(json.dumps({'shape': list(f.shape), 'sum': float(f.double().sum()), 'tokens': m.num_tokens,\n"
            "                  'samples': m.num_samples, 'dim': m.hidden_dim, 'component': m.component}))\n")
        env = dict(os.environ, PYTHONPATH=str(REFERENCE_SRC), PYTHONDONTWRITEBYTECODE="1")
        out = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=env, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        got = json.loads(out.stdout.strip().splitlines()[-1])
        assert got["shape"] == [40, 384] and got["tokens"] == 40 and got["samples"] == 5 and got["dim"] == 384
        assert got["component"] == "decoder"
        assert got["sum"] == float(feats.double().sum())
