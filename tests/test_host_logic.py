"""Host-side logic of the drop-in modules, runnable without a GPU: config schema, module surface,
optimizer/scheduler plumbing, lazy metrics, checkpoints, cache format, loader arithmetic, and the
loud failure on CPU tensors (there is no CPU compute path).  Mirrors the CPU-checkable parts of the
reference's tests/test_config.py, tests/test_sae_model.py and tests/test_training.py."""

from __future__ import annotations

import json

import numpy as np
import pytest
import torch

from whisper_sae import _native as N
from whisper_sae.config import (DataConfig, ExperimentConfig, LayerConfig, SAEConfig, TrainingConfig, WandbConfig,
                                WhisperConfig)
from whisper_sae.sae.model import ReLUSAE, SAEOutput, TopKSAE, create_sae
from whisper_sae.sae.training import SAETrainer, TrainingMetrics, _PendingMetrics


class TestConfig:
    def test_whisper_geometry(self):
        for name, dims in {"tiny": (384, 4), "base": (512, 6), "small": (768, 12), "medium": (1024, 24),
                           "large-v3": (1280, 32)}.items():
            c = WhisperConfig(model_name=f"openai/whisper-{name}")
            assert (c.hidden_dim, c.num_encoder_layers) == dims
        assert WhisperConfig(model_name="custom/model").hidden_dim == 384

    def test_sae_defaults_and_bounds(self):
        c = SAEConfig()
        assert (c.expansion_factor, c.activation, c.k, c.normalize_decoder, c.dead_feature_threshold) == \
            (8, "topk", 32, True, 10_000)
        assert c.get_hidden_dim(384) == 3072 and c.sparsity_weight == 0.01
        for bad in (dict(expansion_factor=2), dict(expansion_factor=64), dict(k=0), dict(activation="tanh")):
            with pytest.raises(ValueError):
                SAEConfig(**bad)

    def test_training_defaults_and_bounds(self):
        c = TrainingConfig()
        assert (c.batch_size, c.learning_rate, c.epochs, c.use_amp, c.seed) == (128, 1e-4, 50, True, 42)
        for bad in (dict(batch_size=0), dict(learning_rate=0), dict(epochs=0)):
            with pytest.raises(ValueError):
                TrainingConfig(**bad)

    def test_yaml_roundtrip_and_reference_yaml_layout(self, tmp_path):
        cfg = ExperimentConfig(whisper=WhisperConfig(model_name="openai/whisper-base"),
                               sae=SAEConfig(expansion_factor=16, k=64), training=TrainingConfig(batch_size=256),
                               experiment_name="rt")
        p = tmp_path / "c.yaml"
        cfg.to_yaml(p)
        back = ExperimentConfig.from_yaml(p)
        assert back.whisper.hidden_dim == 512 and back.sae.k == 64 and back.training.batch_size == 256
        # configs/tiny_test.yaml carries the reference's cfg-1 values (SURVEY.md scope contract)
        from pathlib import Path
        tiny = ExperimentConfig.from_yaml(Path(__file__).parents[1] / "configs" / "tiny_test.yaml")
        assert tiny.sae.dead_feature_threshold == 1000 and tiny.training.batch_size == 64
        assert (tiny.training.epochs, tiny.training.warmup_steps, tiny.training.checkpoint_every) == (3, 100, 2)
        assert tiny.encoder_layers == [0] and tiny.decoder_layers == [] and tiny.wandb.enabled is False
        assert isinstance(tiny.data.cache_dir, Path) and tiny.data.max_samples == 500

    def test_layer_config_and_misc(self, tmp_path):
        lc = LayerConfig(component="decoder", layer_idx=3, input_dim=512, sae_config=SAEConfig(expansion_factor=16))
        assert lc.name == "decoder_layer3" and lc.hidden_dim == 8192
        assert WandbConfig().tags == [] and DataConfig().max_samples == 100_000
        run = ExperimentConfig(output_dir=tmp_path, experiment_name="x").get_run_dir()
        assert run.exists() and run == tmp_path / "x"


class TestModuleSurface:
    def test_attributes_and_state_dict(self, golden_dir):
        api = json.loads((golden_dir / "g9_api.json").read_text())
        m = TopKSAE(384, 3072, k=32, dead_feature_threshold=1000)
        assert (m.input_dim, m.hidden_dim, m.k) == (384, 3072, 32)
        assert (m.encoder.in_features, m.encoder.out_features) == (384, 3072)
        assert (m.decoder.in_features, m.decoder.out_features) == (3072, 384)
        sd = m.state_dict()
        assert list(sd.keys()) == api["state_dict_keys"]
        assert {k: list(v.shape) for k, v in sd.items()} == api["state_dict_shapes"]
        assert {k: str(v.dtype) for k, v in sd.items()} == api["state_dict_dtypes"]
        assert [n for n, _ in m.named_parameters()] == ["b_pre", "encoder.weight", "encoder.bias", "decoder.weight",
                                                        "decoder.bias"]

    def test_decoder_init_column_norm(self):
        m = TopKSAE(384, 3072, k=32)
        cn = m.decoder.weight.data.norm(dim=0)
        assert torch.allclose(cn, torch.full_like(cn, 0.1), atol=1e-5)  # ref test_sae_model.py:55-65
        assert torch.all(m.b_pre == 0) and torch.all(m.feature_last_activated == 0) and m.step_count.item() == 0

    def test_same_seed_same_weights(self):
        torch.manual_seed(7)
        a = TopKSAE(64, 256, k=8)
        torch.manual_seed(7)
        b = TopKSAE(64, 256, k=8)
        assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))

    def test_factory(self):
        assert isinstance(create_sae(SAEConfig(activation="topk", k=16), 384), TopKSAE)
        r = create_sae(SAEConfig(activation="relu", sparsity_weight=0.05), 384)
        assert isinstance(r, ReLUSAE) and r.sparsity_weight == 0.05 and r.hidden_dim == 3072
        assert isinstance(create_sae(SAEConfig(activation="gelu"), 384), ReLUSAE)
        assert SAEOutput._fields == ("reconstructed", "hidden", "loss", "reconstruction_loss", "sparsity_loss", "l0")

    def test_cpu_tensors_fail_loudly(self):
        m = TopKSAE(64, 256, k=8)
        for call in (lambda: m(torch.randn(4, 64)), lambda: m.encode(torch.randn(4, 64)),
                     lambda: m.get_dead_feature_ratio(), lambda: m.normalize_decoder_weights()):
            with pytest.raises(N.WsaeError, match="no CPU path"):
                call()


class TestTrainerHostLogic:
    @pytest.fixture
    def trainer(self, tmp_path):
        m = TopKSAE(64, 128, k=8)
        cfg = TrainingConfig(batch_size=16, learning_rate=1e-3, epochs=2, warmup_steps=10, gradient_clip=1.0,
                             use_amp=False, checkpoint_every=1)
        return SAETrainer(m, cfg, device="cpu", run_dir=tmp_path / "run"), m

    def test_initial_state(self, trainer, tmp_path):
        tr, m = trainer
        assert tr.global_step == 0 and tr.epoch == 0 and tr.model is m and tr.metrics_history == []
        assert (tmp_path / "run").exists() and tr.use_amp is False and tr.scheduler is None
        assert tr._resample_dataset is None and tr.optimizer.param_groups[0]["lr"] == 1e-3
        assert not tr.scaler.is_enabled()

    def test_amp_only_on_gpu(self, tmp_path):
        tr = SAETrainer(TopKSAE(64, 128, k=8), TrainingConfig(use_amp=True), device="cpu", run_dir=tmp_path)
        assert tr.use_amp is False  # ref test_training.py:402-431

    def test_lr_schedule_matches_reference_goldens(self, golden_dir, tmp_path):
        cases = json.loads((golden_dir / "g5_lr_schedule.json").read_text())
        for name, c in cases.items():
            tr = SAETrainer(TopKSAE(32, 64, k=4), TrainingConfig(learning_rate=c["lr"], warmup_steps=c["warmup_cfg"],
                                                                 use_amp=False), device="cpu", run_dir=tmp_path)
            tr.setup_scheduler(c["total"])
            tr.optimizer._opt_called = True  # schedule only; silences torch's step-order warning
            got = []
            for _ in range(len(c["values"])):
                got.append(tr.optimizer.param_groups[0]["lr"])
                tr.scheduler.step()
            assert got == c["values"], name  # same torch scheduler classes -> bit-identical

    def test_lean_scheduler_step_is_torchs_step(self, golden_dir, tmp_path):
        """``SAETrainer._scheduler_step`` (what ``train_step`` calls) against golden set G5 AND against plain ``.step()`` on a
        twin: the rate, ``get_last_lr()`` and the whole ``state_dict()`` stay bit-equal at every step, and a run may switch
        between the two paths at any point."""
        cases = json.loads((golden_dir / "g5_lr_schedule.json").read_text())
        for name, c in cases.items():
            cfg = TrainingConfig(learning_rate=c["lr"], warmup_steps=c["warmup_cfg"], use_amp=False, num_workers=0)
            fast = SAETrainer(TopKSAE(32, 64, k=4), cfg, device="cpu", run_dir=tmp_path / (name + "f"))
            twin = SAETrainer(TopKSAE(32, 64, k=4), cfg, device="cpu", run_dir=tmp_path / (name + "t"))
            fast.setup_scheduler(c["total"])
            twin.setup_scheduler(c["total"])
            assert fast._sched_fast
            for i in range(len(c["values"])):
                assert fast.optimizer.param_groups[0]["lr"] == c["values"][i], (name, i)
                if i % 7 == 3:
                    fast.scheduler.step()       # a plain step in the middle of lean ones
                else:
                    fast._scheduler_step()
                twin.scheduler.step()
                assert fast.optimizer.param_groups[0]["lr"] == twin.optimizer.param_groups[0]["lr"], (name, i)
                assert fast.scheduler.get_last_lr() == twin.scheduler.get_last_lr()
            assert fast.scheduler.state_dict() == twin.scheduler.state_dict(), name

    def test_optimizer_state_dict_layout(self, trainer, golden_dir):
        api = json.loads((golden_dir / "g9_api.json").read_text())
        tr, _ = trainer
        sd = tr.optimizer.state_dict()
        assert sorted(sd["param_groups"][0].keys()) == [k for k in api["optimizer_param_group_keys"] if k != "initial_lr"]
        assert sd["param_groups"][0]["params"] == [0, 1, 2, 3, 4] and sd["state"] == {}

    def test_checkpoint_roundtrip(self, trainer, golden_dir):
        api = json.loads((golden_dir / "g9_api.json").read_text())
        tr, m = trainer
        tr.setup_scheduler(100)
        tr.global_step, tr.epoch = 37, 2
        path = tr.save_checkpoint("c.pt")
        ck = torch.load(path, weights_only=False)
        assert sorted(ck.keys()) == api["checkpoint_keys"]
        tr2 = SAETrainer(TopKSAE(64, 128, k=8), tr.config, device="cpu", run_dir=tr.run_dir)
        tr2.setup_scheduler(100)
        tr2.load_checkpoint(path)
        assert (tr2.global_step, tr2.epoch) == (37, 2)
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), tr2.model.state_dict().values()))

    def test_metrics_json(self, trainer):
        tr, _ = trainer
        tr.metrics_history = [TrainingMetrics(0.5, 0.5, 0.0, 8.0, 0.1, 1e-3, i + 1) for i in range(3)]
        rows = json.loads(tr.save_metrics().read_text())
        assert len(rows) == 3 and rows[0]["step"] == 1 and set(rows[0]) == {
            "step", "loss", "reconstruction_loss", "sparsity_loss", "l0", "dead_feature_ratio", "learning_rate"}

    def test_train_step_on_cpu_raises(self, trainer):
        tr, _ = trainer
        for batch in (torch.randn(16, 64), (torch.randn(16, 64),), [torch.randn(16, 64)]):
            with pytest.raises(N.WsaeError, match="no CPU path"):
                tr.train_step(batch)

    def test_resample_gate(self, trainer):
        tr, _ = trainer
        assert tr._maybe_resample_dead_features() == 0  # no dataset
        tr.set_resample_dataset(torch.utils.data.TensorDataset(torch.randn(10, 64)))
        assert tr._resample_dataset is not None
        tr.global_step = 0
        assert tr._maybe_resample_dead_features() == 0  # step 0 never resamples
        tr.global_step = 7
        assert tr._maybe_resample_dead_features() == 0  # not a multiple of resample_dead_every


class TestLazyMetrics:
    def test_fields_resolve_once_from_the_record(self):
        from whisper_sae.sae.training import _MetricsRing
        ring = _MetricsRing(chunk=2)
        calls = []

        class Stream:
            def synchronize(self):
                calls.append(1)

        chunk, slot = ring.next("cpu")
        chunk.dev[slot].view(torch.float32)[:5] = torch.tensor([0.25, 8.0, 3.0, 0.5, 0.125])
        m = _PendingMetrics(chunk, slot, Stream(), 1e-3, 5)
        chunk2, slot2 = ring.next("cpu")
        assert chunk2 is chunk and slot2 == 1
        chunk.dev[slot2].view(torch.float32)[:2] = torch.tensor([0.5, 4.0])
        m2 = _PendingMetrics(chunk2, slot2, Stream(), 1e-3, 6)
        assert isinstance(m, TrainingMetrics) and m.step == 5 and m.learning_rate == 1e-3 and calls == []
        assert m.loss == 0.25 and calls == [1]
        assert (m.reconstruction_loss, m.sparsity_loss, m.l0, m.dead_feature_ratio) == (0.25, 0.0, 8.0, 0.125)
        assert (m.grad_norm, m.clip_coef) == (3.0, 0.5) and calls == [1]
        assert (m2.loss, m2.l0) == (0.5, 4.0) and calls == [1]  # fetched together with the first record
        chunk3, slot3 = ring.next("cpu")
        assert chunk3 is not chunk and slot3 == 0  # a fresh zeroed chunk once the first is used up
        assert TrainingMetrics(0.5, 0.4, 0.1, 32.0, 0.1, 1e-4, 100).loss == 0.5


class TestCacheFormat:
    def test_save_load_roundtrip_and_names(self, tmp_path):
        from whisper_sae.data.feature_cache import CacheMetadata, FeatureCache
        fc = FeatureCache(tmp_path / "features", WhisperConfig(), DataConfig(cache_dir=tmp_path))
        assert not fc.has_cache("encoder", 0)
        feats = torch.randn(30, 384)
        fc.save(feats, "encoder", 0, num_samples=2)
        assert (tmp_path / "features" / "whisper-tiny_encoder_layer0.pt").exists()
        assert (tmp_path / "features" / "whisper-tiny_encoder_layer0_meta.json").exists()
        assert fc.has_cache("encoder", 0)
        back, meta = fc.load("encoder", 0)
        assert torch.equal(back, feats) and isinstance(meta, CacheMetadata)
        assert (meta.num_tokens, meta.hidden_dim, meta.num_samples, meta.component) == (30, 384, 2, "encoder")
        assert CacheMetadata.from_json(meta.to_json()) == meta and isinstance(meta.data_config["cache_dir"], str)

    def test_ring_loader_arithmetic(self):
        from whisper_sae.data.feature_cache import RingLoader

        class FakeRing:
            device = torch.device("cpu")
            data = torch.zeros(1000, 4)
            launches = 0

            def __len__(self):
                return 1000

            def sample(self, n, seed, epoch, offset):  # identity "permutation": row index = position, tagged with the epoch
                FakeRing.launches += 1
                return torch.arange(offset, offset + n, dtype=torch.int32) + 10000 * epoch

            def batch(self, n, seed, epoch, offset):
                from whisper_sae.sae.training import RingBatch
                return RingBatch(self.data, self.sample(n, seed, epoch, offset))

        ld = RingLoader(FakeRing(), 64)
        assert len(ld) == 16
        got = list(ld)
        assert [len(g) for g in got] == [64] * 15 + [40]
        assert [int(g.rows[0]) for g in got][:3] == [0, 64, 128] and int(got[-1].rows[-1]) == 999
        assert torch.equal(torch.cat([g.rows for g in got]), torch.arange(1000, dtype=torch.int32))  # every row once
        assert FakeRing.launches == 2  # 15 full steps in one launch + the ragged tail
        assert [int(g.rows[0]) // 10000 for g in list(ld)] == [1] * 16  # next epoch reshuffles
        two = RingLoader(FakeRing(), 50, rank=1, world_size=2)
        assert len(two) == 10 and [int(g.rows[0]) for g in two][:2] == [50, 150]
