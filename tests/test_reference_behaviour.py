"""The reference's own behavioural tests, re-expressed against this build on the device.

`tests/test_sae_model.py` and `tests/test_training.py` of omarkhursheed/whisper-sae pin the public behaviour
of `TopKSAE`, `ReLUSAE`, `create_sae` and `SAETrainer` with property checks (shapes, exactly-k sparsity, the
selected set equals `topk`, loss == MSE, unit-norm decoder columns, dead-feature bookkeeping, checkpoint keys,
metrics files ...).  A drop-in must pass the same checks, so each test below carries the name of the
reference test it restates (file:line in the docstring) and differs only in where the tensors live: this
build has no CPU path.  (The reference's three device tests - cpu / no-amp-on-cpu / no-amp-on-mps,
test_training.py:391-430 - have no counterpart for the same reason.)
"""

from __future__ import annotations

import json

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda", 0)


@pytest.fixture
def sae(device):
    from whisper_sae.sae.model import TopKSAE
    return TopKSAE(input_dim=384, hidden_dim=3072, k=32, normalize_decoder=True, dead_feature_threshold=10_000).to(device)


@pytest.fixture
def small_sae(device):
    from whisper_sae.sae.model import TopKSAE
    return TopKSAE(input_dim=64, hidden_dim=256, k=8, normalize_decoder=True, dead_feature_threshold=100).to(device)


class TestTopKSAE:
    def test_initialization(self, sae):
        """test_sae_model.py:44"""
        assert (sae.input_dim, sae.hidden_dim, sae.k) == (384, 3072, 32)
        assert (sae.encoder.in_features, sae.encoder.out_features) == (384, 3072)
        assert (sae.decoder.in_features, sae.decoder.out_features) == (3072, 384)
        assert sae.b_pre.shape == (384,)

    def test_decoder_initialization_normalized(self, sae):
        """test_sae_model.py:55 - columns of norm 0.1 after the x0.1 scaling of model.py:89"""
        norms = sae.decoder.weight.data.norm(dim=0)
        assert torch.allclose(norms, torch.full_like(norms, 0.1), atol=1e-5)

    def test_normalize_decoder_weights(self, sae):
        """test_sae_model.py:67"""
        with torch.no_grad():
            sae.decoder.weight.mul_(3.7)
        sae.normalize_decoder_weights()
        norms = sae.decoder.weight.data.norm(dim=0)
        assert torch.allclose(norms, torch.ones_like(norms), atol=1e-5)

    def test_encode_output_shape(self, sae, device):
        """test_sae_model.py:79"""
        assert sae.encode(torch.randn(16, 384, device=device)).shape == (16, 3072)

    def test_topk_sparsity(self, sae, device):
        """test_sae_model.py:86 - at most k non-zeros per row"""
        hidden = sae.encode(torch.randn(16, 384, device=device))
        assert torch.all((hidden != 0).sum(dim=-1) <= sae.k)

    def test_topk_values_are_positive(self, sae, device):
        """test_sae_model.py:99"""
        assert torch.all(sae.encode(torch.randn(16, 384, device=device)) >= 0)

    def test_topk_selects_largest(self, sae, device):
        """test_sae_model.py:110 - the non-zero positions are the positive members of topk(pre)"""
        x = torch.randn(8, 384, device=device)
        pre = sae.pre_activation(x)
        hidden = sae.encode(x)
        vals, idx = torch.topk(pre, sae.k, dim=-1)
        for b in range(x.shape[0]):
            want = set(idx[b][vals[b] > 0].tolist())
            assert set(torch.nonzero(hidden[b]).flatten().tolist()) == want

    def test_decode_output_shape(self, sae, device):
        """test_sae_model.py:132"""
        assert sae.decode(torch.randn(16, 3072, device=device)).shape == (16, 384)

    def test_forward_returns_sae_output(self, sae, device):
        """test_sae_model.py:139"""
        from whisper_sae.sae.model import SAEOutput
        out = sae(torch.randn(16, 384, device=device))
        assert isinstance(out, SAEOutput)
        assert out._fields == ("reconstructed", "hidden", "loss", "reconstruction_loss", "sparsity_loss", "l0")

    def test_forward_output_shapes(self, sae, device):
        """test_sae_model.py:152"""
        out = sae(torch.randn(16, 384, device=device))
        assert out.reconstructed.shape == (16, 384) and out.hidden.shape == (16, 3072)
        assert out.loss.ndim == 0 and out.reconstruction_loss.ndim == 0 and out.l0.ndim == 0

    def test_reconstruction_loss_is_mse(self, sae, device):
        """test_sae_model.py:164"""
        x = torch.randn(16, 384, device=device)
        out = sae(x)
        assert torch.isclose(out.reconstruction_loss, torch.nn.functional.mse_loss(out.reconstructed, x), rtol=1e-5)

    def test_sparsity_loss_is_zero_for_topk(self, sae, device):
        """test_sae_model.py:174"""
        assert sae(torch.randn(16, 384, device=device)).sparsity_loss.item() == 0.0

    def test_l0_equals_k(self, sae, device):
        """test_sae_model.py:181 (l0 <= k: negative winners are zeroed)"""
        out = sae(torch.randn(16, 384, device=device))
        assert out.l0.item() <= sae.k and torch.isclose(out.l0.cpu(), torch.tensor(float(sae.k)), atol=0.1 * sae.k)

    def test_dead_feature_tracking_initialization(self, sae):
        """test_sae_model.py:188"""
        assert sae.feature_last_activated.shape == (sae.hidden_dim,)
        assert torch.all(sae.feature_last_activated == 0) and sae.step_count.item() == 0

    def test_dead_feature_tracking_updates(self, small_sae, device):
        """test_sae_model.py:194"""
        small_sae.train()
        small_sae(torch.randn(32, 64, device=device))
        assert small_sae.step_count.item() == 1
        assert (small_sae.feature_last_activated > 0).sum().item() > 0

    def test_dead_feature_tracking_not_updated_in_eval(self, small_sae, device):
        """test_sae_model.py:208"""
        small_sae.eval()
        small_sae(torch.randn(32, 64, device=device))
        assert small_sae.step_count.item() == 0

    def test_get_dead_features_initially_all_dead(self, small_sae):
        """test_sae_model.py:217 (despite its name: initially NO feature is dead)"""
        assert torch.all(~small_sae.get_dead_features())

    def test_get_dead_features_after_many_steps(self, small_sae, device):
        """test_sae_model.py:227"""
        small_sae.train()
        torch.manual_seed(12345)
        fixed = torch.randn(8, 64, device=device)
        for _ in range(150):
            small_sae(fixed)
        mask = small_sae.get_dead_features()
        assert mask.shape == (256,) and small_sae.step_count.item() == 150
        # features never selected by the fixed batch have been silent for 150 > 100 steps
        assert mask.sum().item() == int((small_sae.feature_last_activated == 0).sum().item()) > 0
        assert 0.0 < small_sae.get_dead_feature_ratio() < 1.0

    def test_gradients_flow(self, small_sae, device):
        """test_sae_model.py:296"""
        small_sae(torch.randn(16, 64, device=device)).loss.backward()
        for p in (small_sae.encoder.weight, small_sae.encoder.bias, small_sae.decoder.weight, small_sae.decoder.bias,
                  small_sae.b_pre):
            assert p.grad is not None and torch.isfinite(p.grad).all()
        assert small_sae.encoder.weight.grad.abs().sum() > 0 and small_sae.decoder.weight.grad.abs().sum() > 0

    def test_deterministic_with_same_input(self, sae, device):
        """test_sae_model.py:311"""
        sae.eval()
        x = torch.randn(8, 384, device=device)
        a, b = sae(x), sae(x)
        assert torch.equal(a.reconstructed, b.reconstructed) and torch.equal(a.hidden, b.hidden)


class TestReLUSAE:
    @pytest.fixture
    def relu(self, device):
        from whisper_sae.sae.model import ReLUSAE
        return ReLUSAE(input_dim=384, hidden_dim=3072, sparsity_weight=0.01).to(device)

    def test_initialization(self, relu):
        """test_sae_model.py:336"""
        assert (relu.input_dim, relu.hidden_dim, relu.sparsity_weight) == (384, 3072, 0.01)

    def test_forward_shapes(self, relu, device):
        """test_sae_model.py:342"""
        out = relu(torch.randn(16, 384, device=device))
        assert out.reconstructed.shape == (16, 384) and out.hidden.shape == (16, 3072)

    def test_sparsity_loss_nonzero(self, relu, device):
        """test_sae_model.py:350"""
        assert relu(torch.randn(16, 384, device=device)).sparsity_loss.item() > 0

    def test_total_loss_includes_sparsity(self, relu, device):
        """test_sae_model.py:358"""
        out = relu(torch.randn(16, 384, device=device))
        want = out.reconstruction_loss + relu.sparsity_weight * out.sparsity_loss
        assert torch.isclose(out.loss.detach(), want, rtol=1e-5)


class TestCreateSAE:
    def test_create_topk_sae(self):
        """test_sae_model.py:370"""
        from whisper_sae.config import SAEConfig
        from whisper_sae.sae.model import TopKSAE, create_sae
        m = create_sae(SAEConfig(activation="topk", k=64, expansion_factor=8), input_dim=384)
        assert isinstance(m, TopKSAE) and (m.k, m.hidden_dim) == (64, 3072)

    def test_create_relu_sae(self):
        """test_sae_model.py:385"""
        from whisper_sae.config import SAEConfig
        from whisper_sae.sae.model import ReLUSAE, create_sae
        m = create_sae(SAEConfig(activation="relu", expansion_factor=4), input_dim=384)
        assert isinstance(m, ReLUSAE) and m.hidden_dim == 1536

    def test_create_sae_with_different_expansions(self):
        """test_sae_model.py:398"""
        from whisper_sae.config import SAEConfig
        from whisper_sae.sae.model import create_sae
        for f in (4, 8, 16, 32):
            assert create_sae(SAEConfig(expansion_factor=f), input_dim=384).hidden_dim == 384 * f


class TestSAEReconstruction:
    def test_reconstruction_uses_k_features(self, device):
        """test_sae_model.py:409"""
        from whisper_sae.sae.model import TopKSAE
        m = TopKSAE(64, 256, k=4).to(device)
        assert torch.all((m(torch.randn(8, 64, device=device)).hidden != 0).sum(-1) <= 4)

    def test_reconstruction_improves_with_training(self, device):
        """test_sae_model.py:426 - plain torch.optim.Adam on the module's parameters through autograd"""
        from whisper_sae.sae.model import TopKSAE
        torch.manual_seed(0)
        m = TopKSAE(64, 256, k=16).to(device)
        opt = torch.optim.Adam(m.parameters(), lr=1e-2)
        data = torch.randn(256, 64, device=device)
        first = m(data).loss.item()
        for _ in range(60):
            opt.zero_grad()
            m(data).loss.backward()
            opt.step()
            m.normalize_decoder_weights()
        assert m(data).loss.item() < 0.8 * first


class TestTrainingMetrics:
    def test_training_metrics_creation(self):
        """test_training.py:18"""
        from whisper_sae.sae.training import TrainingMetrics
        t = TrainingMetrics(loss=0.5, reconstruction_loss=0.4, sparsity_loss=0.1, l0=32.0, dead_feature_ratio=0.1,
                            learning_rate=1e-4, step=100)
        assert (t.loss, t.reconstruction_loss, t.sparsity_loss, t.l0, t.dead_feature_ratio, t.learning_rate, t.step) == \
            (0.5, 0.4, 0.1, 32.0, 0.1, 1e-4, 100)


@pytest.fixture
def training_config():
    from whisper_sae.config import TrainingConfig
    return TrainingConfig(batch_size=32, learning_rate=1e-3, weight_decay=0.0, epochs=2, warmup_steps=10, gradient_clip=1.0,
                          use_amp=False, checkpoint_every=1, num_workers=0)


@pytest.fixture
def simple_model(device):
    from whisper_sae.sae.model import TopKSAE
    return TopKSAE(64, 256, k=8, dead_feature_threshold=100)


@pytest.fixture
def sample_loader(device):
    data = torch.randn(128, 64)
    return torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=32)


class TestSAETrainer:
    def _trainer(self, model, cfg, device, tmp_path, **kw):
        from whisper_sae.sae.training import SAETrainer
        return SAETrainer(model, cfg, device=device, run_dir=tmp_path / "run", **kw)

    def test_trainer_initialization(self, simple_model, training_config, device, tmp_path):
        """test_training.py:68"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        assert tr.model is simple_model and tr.config is training_config
        assert tr.global_step == 0 and tr.epoch == 0 and tr.metrics_history == []

    def test_trainer_with_run_dir(self, simple_model, training_config, device, tmp_path):
        """test_training.py:80"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        assert tr.run_dir == tmp_path / "run" and tr.run_dir.exists()

    def test_setup_scheduler(self, simple_model, training_config, device, tmp_path):
        """test_training.py:92"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        assert tr.scheduler is None
        tr.setup_scheduler(total_steps=1000)
        assert tr.scheduler is not None

    def test_train_step_returns_metrics(self, simple_model, training_config, device, tmp_path):
        """test_training.py:102"""
        from whisper_sae.sae.training import TrainingMetrics
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        m = tr.train_step(torch.randn(32, 64))
        assert isinstance(m, TrainingMetrics) and m.loss > 0 and m.reconstruction_loss > 0
        assert m.l0 <= 8 and m.step == 1 and m.learning_rate == training_config.learning_rate

    def test_train_step_handles_tuple_batch(self, simple_model, training_config, device, tmp_path):
        """test_training.py:120"""
        assert self._trainer(simple_model, training_config, device, tmp_path).train_step((torch.randn(32, 64),)).loss > 0

    def test_train_step_handles_list_batch(self, simple_model, training_config, device, tmp_path):
        """test_training.py:136"""
        assert self._trainer(simple_model, training_config, device, tmp_path).train_step([torch.randn(32, 64)]).loss > 0

    def test_train_step_increments_global_step(self, simple_model, training_config, device, tmp_path):
        """test_training.py:150"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        for i in range(3):
            tr.train_step(torch.randn(32, 64))
            assert tr.global_step == i + 1

    def test_train_step_updates_scheduler(self, simple_model, training_config, device, tmp_path):
        """test_training.py:165"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        tr.setup_scheduler(total_steps=100)
        lr0 = tr.optimizer.param_groups[0]["lr"]
        tr.train_step(torch.randn(32, 64))
        assert tr.optimizer.param_groups[0]["lr"] != lr0

    def test_train_epoch(self, simple_model, training_config, device, tmp_path, sample_loader):
        """test_training.py:184"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        ms = tr.train_epoch(sample_loader)
        assert len(ms) == 4 and tr.epoch == 1 and tr.global_step == 4

    def test_train_epoch_records_metrics(self, simple_model, training_config, device, tmp_path, sample_loader):
        """test_training.py:200"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        tr.train_epoch(sample_loader)
        assert len(tr.metrics_history) == 4 and all(m.loss > 0 for m in tr.metrics_history)

    def test_loss_decreases_during_training(self, simple_model, training_config, device, tmp_path):
        """test_training.py:214"""
        torch.manual_seed(42)
        data = torch.randn(256, 64)
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=32)
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        first = tr.train_epoch(loader)
        for _ in range(4):
            lastm = tr.train_epoch(loader)
        assert sum(m.loss for m in lastm) < sum(m.loss for m in first)

    def test_save_checkpoint(self, simple_model, training_config, device, tmp_path):
        """test_training.py:242"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        tr.train_step(torch.randn(32, 64))
        tr.save_checkpoint("test_checkpoint.pt")
        path = tr.run_dir / "test_checkpoint.pt"
        assert path.exists()
        ck = torch.load(path, weights_only=False)
        for key in ("model_state_dict", "optimizer_state_dict", "global_step", "epoch", "config"):
            assert key in ck
        assert ck["global_step"] == 1
        assert set(ck["model_state_dict"]) == {"b_pre", "encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias",
                                               "feature_last_activated", "step_count"}
        assert ck["model_state_dict"]["decoder.weight"].shape == (64, 256)

    def test_load_checkpoint(self, simple_model, training_config, device, tmp_path):
        """test_training.py:263"""
        from whisper_sae.sae.model import TopKSAE
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        for _ in range(5):
            tr.train_step(torch.randn(32, 64))
        tr.save_checkpoint("ck.pt")
        fresh = self._trainer(TopKSAE(64, 256, k=8, dead_feature_threshold=100), training_config, device, tmp_path)
        fresh.load_checkpoint(tr.run_dir / "ck.pt")
        assert fresh.global_step == 5
        for (ka, a), (kb, b) in zip(sorted(tr.model.state_dict().items()), sorted(fresh.model.state_dict().items())):
            assert ka == kb and torch.equal(a.cpu(), b.cpu()), ka
        x = torch.randn(32, 64)  # identical continuation: same optimizer state, same step
        assert tr.train_step(x).loss == fresh.train_step(x).loss

    def test_save_metrics(self, simple_model, training_config, device, tmp_path, sample_loader):
        """test_training.py:292"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        tr.train_epoch(sample_loader)
        tr.save_metrics()
        rows = json.loads((tr.run_dir / "metrics.json").read_text())
        assert len(rows) == 4 and {"loss", "l0", "step", "learning_rate", "dead_feature_ratio"} <= set(rows[0])

    def test_decoder_normalization(self, simple_model, training_config, device, tmp_path):
        """test_training.py:314"""
        tr = self._trainer(simple_model, training_config, device, tmp_path)
        for _ in range(5):
            tr.train_step(torch.randn(32, 64))
        norms = simple_model.decoder.weight.data.norm(dim=0)
        assert torch.allclose(norms, torch.ones_like(norms), atol=1e-5)


class TestTrainerResampling:
    def test_set_resample_dataset(self, simple_model, training_config, device, tmp_path):
        """test_training.py:352"""
        from whisper_sae.sae.training import SAETrainer
        tr = SAETrainer(simple_model, training_config, device=device, run_dir=tmp_path, resample_dead_every=10,
                        resample_batch_size=64)
        ds = torch.utils.data.TensorDataset(torch.randn(100, 64))
        tr.set_resample_dataset(ds)
        assert tr._resample_dataset is ds

    def test_resampling_triggers_at_interval(self, device, training_config, tmp_path):
        """test_training.py:367 - with the call wired in (resample_dead=True; the reference never calls it)"""
        from whisper_sae.sae.model import TopKSAE
        from whisper_sae.sae.training import SAETrainer
        m = TopKSAE(64, 256, k=8, dead_feature_threshold=5)
        tr = SAETrainer(m, training_config, device=device, run_dir=tmp_path, resample_dead_every=10, resample_batch_size=64,
                        resample_dead=True)
        tr.set_resample_dataset(torch.utils.data.TensorDataset(torch.randn(100, 64)))
        torch.manual_seed(1)
        fixed = torch.randn(32, 64)
        for _ in range(10):
            tr.train_step(fixed)
        assert tr.global_step == 10 and tr.num_resampled_total > 0
