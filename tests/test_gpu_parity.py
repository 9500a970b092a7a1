"""GPU parity: the HIP path (through the C ABI, via the drop-in modules) against the CPU oracle and
the committed golden vectors of the real reference.  Run on the MI355X box: ``pytest -m gpu``.

Tolerances (stated per assertion):
* TopK index sets: bit-exact (fixtures carry an asserted k/k+1 margin);
* reconstruction / loss: 1e-5 relative (north_star);
* gradients, fp32 mode: 2e-5 relative to the tensor's max; bf16 mode: 1e-4 against the oracle's
  bf16-mirrored arithmetic, 2e-2 against the fp32 reference values;
* parameters after a step: absolute 2e-7 on O(lr) updates.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu

from pathlib import Path  # noqa: E402

GOLDEN_ROOT = Path(__file__).resolve().parents[1]

KEYS = ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias", "b_pre")

# multi-step tolerances = measured on the MI355X x 3 (profiles/r03_parity_notes.jsonl holds the measured values: the fp32
# mode's exact-fp32 MFMA is a k-ordered fmaf chain and tracks torch's CPU trajectory to ~1e-7 over 20 steps, two decades
# inside north_star's 1e-5; round 2 carried 5e-5 / 2e-4 / 2e-6 here without saying how much of that was slack)
G3_WD_ABS = 2e-7      # measured 5.2e-8 (the same bound as every other tensor of the one-step pin)
G4_LOSS_REL = 7e-7    # measured 2.2e-7
G4_STATE_REL = 2.5e-6  # measured 7.8e-7 (parameters)
G4_MOMENT_REL = 4e-5  # measured 1.3e-5 (exp_avg_sq of entries whose gradient is near zero)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def build(D, H, k, seed, bf16, b_pre_scale, thr, device, precision):
    from whisper_sae.sae.model import TopKSAE
    w = synth.sae_weights(D, H, seed=seed, bf16=bf16, b_pre_scale=b_pre_scale)
    m = TopKSAE(D, H, k=k, dead_feature_threshold=thr, precision=precision)
    sd = m.state_dict()
    for key in KEYS:
        sd[key] = torch.from_numpy(w[key])
    m.load_state_dict(sd)
    m.to(device)
    st = O.SAEState.from_state_dict(w, k=k, dead_feature_threshold=thr)
    return m, st


def cpu(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(golden_dir / "g1_forward_cfg2.npz")


class TestForwardCfg2:
    @pytest.mark.parametrize("precision,mode", [("fp32", "fp32"), ("bf16", "amp")])
    def test_golden_forward(self, g1, device, precision, mode):
        D, H, K, B = (int(v) for v in g1["dims"])
        m, st = build(D, H, K, 42, True, 0.1, 1000, device, precision)
        x = synth.activations(B, D, seed=42, stream=1, bf16=True)
        m.train()
        out = m(torch.from_numpy(x).to(device))
        vals, idx = m._last_code
        # TopK index sets bit-exact vs the real reference (and vs the oracle)
        got = np.sort(cpu(idx).astype(np.int64), axis=1)
        assert np.array_equal(got, np.sort(g1["idx"].astype(np.int64), axis=1))
        ora = O.forward(st.copy(), x, mode)
        assert np.array_equal(got, np.sort(ora["idx"], axis=1))
        assert rel(cpu(vals), g1["vals"]) < 1e-5
        # reconstruction and loss within 1e-5 relative of the reference
        assert rel(cpu(out.reconstructed), g1["recon"]) < 1e-5
        assert abs(out.loss.item() - float(g1["loss"])) / float(g1["loss"]) < 1e-5
        assert out.l0.item() == float(g1["l0"])
        assert float(out.sparsity_loss) == 0.0
        # hidden is the dense scatter of relu(topk)
        hid = cpu(out.hidden)
        assert hid.shape == (B, H)
        assert np.array_equal((hid != 0).sum(1), np.full(B, K))
        assert rel(hid, ora["hidden"]) < 1e-5
        # dead-feature clock
        assert int(m.step_count.item()) == int(g1["step_count"]) == 1
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), g1["last_activated"])

    def test_eval_mode_leaves_clock(self, g1, device):
        D, H, K, B = (int(v) for v in g1["dims"])
        m, _ = build(D, H, K, 42, True, 0.1, 1000, device, "fp32")
        m.eval()
        x = torch.from_numpy(synth.activations(B, D, seed=42, stream=1)).to(device)
        a = m(x)
        b = m(x)
        assert int(m.step_count.item()) == 0
        assert torch.equal(a.reconstructed, b.reconstructed)  # eval determinism (ref test :311-320)

    def test_dense_pre_activation_and_api_paths(self, g1, device):
        D, H, K, B = (int(v) for v in g1["dims"])
        m, st = build(D, H, K, 42, True, 0.1, 1000, device, "fp32")
        x = synth.activations(B, D, seed=42, stream=1)
        xt = torch.from_numpy(x).to(device)
        pre = cpu(m.pre_activation(xt))
        assert rel(pre, O.pre_activation(st, x, "fp32")) < 1e-5
        hidden = m.encode(xt)
        rec = m.decode(hidden)
        assert rel(cpu(rec), g1["recon"]) < 1e-5
        dense = torch.from_numpy(synth.normal((8, H), 3, 9)).to(device)
        assert rel(cpu(m.decode(dense)), O.decode(st, synth.normal((8, H), 3, 9))) < 1e-5


class TestGradsCfg2:
    @pytest.mark.parametrize("precision,mode,tol_ref,tol_ora", [("fp32", "fp32", 2e-5, 2e-5), ("bf16", "amp", 2e-2, 2e-4)])
    def test_backward(self, g1, golden_dir, device, precision, mode, tol_ref, tol_ora):
        g2 = np.load(golden_dir / "g2_grads_cfg2.npz")
        D, H, K, B = (int(v) for v in g1["dims"])
        m, st = build(D, H, K, 42, True, 0.1, 1000, device, precision)
        x = synth.activations(B, D, seed=42, stream=1, bf16=True)
        xt = torch.from_numpy(x).to(device).requires_grad_(True)
        out = m(xt)
        out.loss.backward()
        got = {"W_e": m.encoder.weight.grad, "b_e": m.encoder.bias.grad, "W_d": m.decoder.weight.grad,
               "b_d": m.decoder.bias.grad, "b_pre": m.b_pre.grad}
        got = {k: cpu(v) for k, v in got.items()}
        assert got["W_d"].shape == (D, H) and got["W_e"].shape == (H, D)
        fwd = O.forward(st.copy(), x, mode)
        ora = O.backward(st, x, fwd, mode)
        for n in ("W_e", "b_e", "W_d", "b_d", "b_pre"):
            assert rel(got[n], ora[n]) < tol_ora, (n, rel(got[n], ora[n]))
        # against the real reference's autograd
        norms = [np.sqrt((got[n].astype(np.float64) ** 2).sum()) for n in ("W_e", "b_e", "W_d", "b_d", "b_pre")]
        assert np.allclose(norms, g2["norms"], rtol=tol_ref)
        for n in ("b_e", "b_d", "b_pre"):
            assert rel(got[n], g2[n]) < tol_ref, n
        assert rel(got["W_e"].reshape(-1)[g2["pos_e"]], g2["W_e_samples"]) < tol_ref * 5
        assert rel(got["W_d"].reshape(-1)[g2["pos_d"]], g2["W_d_samples"]) < tol_ref * 5
        # input gradient: dL/dx = dpre W_e - g   (fp32 oracle)
        if mode == "fp32":
            hidden_mask = fwd["hidden"] > 0
            g64 = ora["g"].astype(np.float64)
            dh = g64 @ st.W_d.astype(np.float64)
            dx = np.where(hidden_mask, dh, 0.0) @ st.W_e.astype(np.float64) - g64
            assert rel(cpu(xt.grad), dx) < 2e-5


class TestShapesAgainstOracle:
    """Forward + backward against the oracle on shapes that leave the tuned fast paths or their tile grids:
    ragged batches (last 64-row chunk / 256-row GEMM tile partly filled; B = 2500 takes the persistent GEMM and
    the strip-guided TopK) and BASELINE.json configs[3] dimensions (768 -> 12288, k = 64: generic decode and
    TopK kernels, two column tiles in the weight-gradient kernel)."""

    @pytest.mark.parametrize("D,H,K,B,precision,mode,tol", [
        (384, 3072, 32, 1000, "fp32", "fp32", 2e-5),
        (384, 3072, 32, 1000, "bf16", "amp", 2e-3),
        (384, 3072, 32, 2500, "bf16", "amp", 2e-3),
        (768, 12288, 64, 192, "fp32", "fp32", 2e-5),
        # bf16 mode feeds the weight-gradient MFMAs bf16-rounded dpre / g / hidden: an element that sits on a
        # rounding boundary lands on the other side when the fp32 sums are taken in a different order (oracle: numpy,
        # kernel: MFMA / wave order), i.e. single entries differ by one bf16 ulp (0.4 %) - hence 2e-3 of the max
        (768, 12288, 64, 192, "bf16", "amp", 2e-3),
        # every width / k class of the MFMA decode kernel (D / 128 column pieces x one or two 32-row gather groups)
        (128, 1024, 16, 300, "bf16", "amp", 2e-3),
        (256, 2048, 32, 300, "bf16", "amp", 2e-3),
        (384, 4096, 64, 200, "bf16", "amp", 2e-3),
        (768, 3072, 32, 200, "bf16", "amp", 2e-3),
        (1280, 5120, 32, 200, "bf16", "amp", 2e-3),
        (1280, 5120, 64, 200, "bf16", "amp", 2e-3),
    ])
    def test_forward_backward(self, device, D, H, K, B, precision, mode, tol):
        m, st = build(D, H, K, 21, True, 0.1, 1000, device, precision)
        x = synth.activations(B, D, seed=21, stream=3, bf16=True)
        m.train()
        out = m(torch.from_numpy(x).to(device))
        out.loss.backward()
        fwd = O.forward(st.copy(), x, mode)
        ora = O.backward(st, x, fwd, mode)
        vals, idx = m._last_code
        # rows whose k-th / (k+1)-th pre-activations are closer than the summation-order noise may legitimately
        # differ in the last selected index: compare the index sets on rows with a clear margin
        clear = synth.topk_margin(fwd["pre"], K) > 1e-5  # relative gap between the k-th and (k+1)-th value
        assert clear.mean() > 0.98
        got_sets = np.sort(idx.cpu().numpy(), axis=1)[clear]
        assert np.array_equal(got_sets, np.sort(fwd["idx"], axis=1)[clear])
        assert abs(float(out.loss.detach()) - float(fwd["loss"])) / float(fwd["loss"]) < max(tol, 1e-5)
        got = {"W_e": m.encoder.weight.grad, "b_e": m.encoder.bias.grad, "W_d": m.decoder.weight.grad,
               "b_d": m.decoder.bias.grad, "b_pre": m.b_pre.grad}
        if clear.all():
            for n, g in got.items():
                assert rel(cpu(g), ora[n]) < tol, (n, rel(cpu(g), ora[n]))
        else:  # a row with a different last index changes a few gradient rows: hold the norms instead
            for n, g in got.items():
                a, b = np.linalg.norm(cpu(g).astype(np.float64)), np.linalg.norm(ora[n].astype(np.float64))
                assert abs(a - b) / b < 1e-3, n


class TestTrainStep:
    def test_g3_one_step_fp32(self, g1, golden_dir, device, tmp_path, parity_note):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer, TrainingMetrics
        g3 = np.load(golden_dir / "g3_train_step_cfg2.npz")
        D, H, K, B = (int(v) for v in g1["dims"])
        m, _ = build(D, H, K, 42, True, 0.1, 1000, "cpu", None)
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-4, weight_decay=0.0, epochs=3, warmup_steps=100,
                             gradient_clip=1.0, use_amp=False, checkpoint_every=2, seed=42, num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        tr.setup_scheduler(35157)
        assert tr.optimizer.param_groups[0]["lr"] == float(g3["lr0"])
        x = torch.from_numpy(synth.activations(B, D, seed=42, stream=1, bf16=True))
        met = tr.train_step(x)
        assert isinstance(met, TrainingMetrics)
        assert met.step == 1 and tr.global_step == 1
        assert abs(met.loss - float(g3["loss"])) / float(g3["loss"]) < 1e-5
        assert met.reconstruction_loss == met.loss and met.sparsity_loss == 0.0
        assert met.l0 == float(g3["l0"])
        assert met.dead_feature_ratio == float(g3["dead_ratio"])
        assert met.learning_rate == float(g3["lr_after"])
        assert abs(met.grad_norm - float(g3["grad_total_norm"])) / float(g3["grad_total_norm"]) < 1e-5
        sd = {k: cpu(v) for k, v in m.state_dict().items()}
        for name, key in (("encoder.bias", "b_e"), ("decoder.bias", "b_d"), ("b_pre", "b_pre")):
            assert np.abs(sd[name] - g3[key]).max() < 2e-7, name
        d_e = np.abs(sd["encoder.weight"].reshape(-1)[g3["pos_e"]] - g3["W_e_samples"]).max()
        d_d = np.abs(sd["decoder.weight"].reshape(-1)[g3["pos_d"]] - g3["W_d_samples"]).max()
        parity_note("g3_We_abs", d_e, 2e-7)
        parity_note("g3_Wd_abs", d_d, G3_WD_ABS)
        assert d_e < 2e-7
        # W_d entries are O(0.1) after the renorm (W_e entries O(0.01)): one fp32 ulp there is 7.5e-9 and the renorm divides by
        # a norm both sides sum in a different order; bound = measured x 3 (parity_notes.jsonl)
        assert d_d < G3_WD_ABS
        cn = np.linalg.norm(sd["decoder.weight"].astype(np.float64), axis=0)
        assert abs(cn.min() - 1) < 1e-5 and abs(cn.max() - 1) < 1e-5  # ref test_training.py:314-326
        # tuple / list batch forms (ref test_training.py:120-148), same losses as the reference's steps 2, 3
        met_t = tr.train_step((x,))
        met_l = tr.train_step([x])
        assert abs(met_t.loss - g3["losses_3steps"][1]) / g3["losses_3steps"][1] < 2e-5
        assert abs(met_l.loss - g3["losses_3steps"][2]) / g3["losses_3steps"][2] < 2e-5
        assert met_l.learning_rate == g3["lrs_3steps"][2]

    def test_g4_trajectory_fp32(self, golden_dir, device, tmp_path, parity_note):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer
        g = np.load(golden_dir / "g4_trajectory_small.npz")
        D, H, K, B, STEPS = (int(v) for v in g["dims"])
        m, _ = build(D, H, K, 7, False, 0.05, 5, "cpu", None)
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.01, epochs=1, warmup_steps=5,
                             gradient_clip=1.0, use_amp=False, num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        tr.setup_scheduler(STEPS)
        xs = synth.activations(B * STEPS, D, seed=7, stream=2, bf16=False).reshape(STEPS, B, D)
        mets = [tr.train_step(torch.from_numpy(xs[s])) for s in range(STEPS)]
        parity_note("g4_loss_rel_fp32", max(abs(met.loss - g["losses"][s]) / g["losses"][s] for s, met in enumerate(mets)), G4_LOSS_REL)
        for s, met in enumerate(mets):
            assert abs(met.loss - g["losses"][s]) / g["losses"][s] < G4_LOSS_REL, s
            assert met.dead_feature_ratio == pytest.approx(g["dead"][s], abs=1e-7), s
        sd = {k: cpu(v) for k, v in m.state_dict().items()}
        parity_note("g4_state_rel_fp32", max(rel(sd["encoder.weight"], g["W_e"]), rel(sd["decoder.weight"], g["W_d"]),
                                             rel(sd["b_pre"], g["b_pre"])), G4_STATE_REL)
        assert rel(sd["encoder.weight"], g["W_e"]) < G4_STATE_REL
        assert rel(sd["decoder.weight"], g["W_d"]) < G4_STATE_REL
        assert rel(sd["b_pre"], g["b_pre"]) < G4_STATE_REL
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), g["last_activated"])
        assert int(m.step_count.item()) == int(g["step_count"])
        osd = tr.optimizer.state_dict()
        parity_note("g4_moments_rel_fp32", max(rel(cpu(osd["state"][1]["exp_avg"]), g["exp_avg_We"]),
                                               rel(cpu(osd["state"][1]["exp_avg_sq"]), g["exp_avg_sq_We"])), G4_MOMENT_REL)
        assert rel(cpu(osd["state"][1]["exp_avg"]), g["exp_avg_We"]) < G4_MOMENT_REL
        assert rel(cpu(osd["state"][1]["exp_avg_sq"]), g["exp_avg_sq_We"]) < G4_MOMENT_REL
        assert float(osd["state"][1]["step"]) == STEPS

    @pytest.mark.parametrize("precision", ["fp32", "bf16"])
    def test_g4b_trajectory_cfg2(self, golden_dir, device, tmp_path, precision, parity_note):
        """The reference's 20 steps at 384 -> 3072, k = 32, B = 512 (G4b) against the product trainer in
        both arithmetic modes.  The oracle's own gap to these numbers (tests/test_oracle_golden.py::TestTrajectoryG4b,
        measured 2.0e-5 fp32 / 3.9e-4 amp) says what two fp32 implementations with different summation orders can
        agree on over 20 Adam steps; the bounds here are measured x 3 and every measured value is written to
        parity_notes.jsonl."""
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer
        g = np.load(golden_dir / "g4b_trajectory_cfg2.npz")
        D, H, K, B, STEPS = (int(v) for v in g["dims"])
        m, _ = build(D, H, K, 11, True, 0.1, 10_000, "cpu", None)
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.0, epochs=1, warmup_steps=5,
                             gradient_clip=1.0, use_amp=(precision == "bf16"), num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        tr.setup_scheduler(STEPS)
        xs = synth.activations(B * STEPS, D, seed=11, stream=4, bf16=True).reshape(STEPS, B, D)
        mets = [tr.train_step(torch.from_numpy(xs[s])) for s in range(STEPS)]
        gap = max(abs(met.loss - g["losses"][s]) / g["losses"][s] for s, met in enumerate(mets))
        for s, met in enumerate(mets):
            assert met.l0 == g["l0"][s] and met.dead_feature_ratio == g["dead"][s], s
            if s + 1 < STEPS:  # the metric carries the rate AFTER the scheduler step = the next step's rate
                assert met.learning_rate == pytest.approx(g["lrs"][s + 1], rel=1e-9), s
        sd = {k: cpu(v) for k, v in m.state_dict().items()}
        drift = max(np.abs(sd[key].reshape(-1)[g[f"pos_{short}"]] - g[f"val_{short}"]).max() / np.abs(g[f"val_{short}"]).max()
                    for key, short in (("encoder.weight", "W_e"), ("decoder.weight", "W_d"), ("encoder.bias", "b_e"),
                                       ("decoder.bias", "b_d"), ("b_pre", "b_pre")))
        # measured on the MI355X: fp32 1.07e-7 / 3.1e-7 (the numpy oracle, with its own summation order, is 2.0e-5 / 2.9e-4
        # away from the same reference numbers); bf16 2.8e-4 / 0.045 - the trajectory gap bf16 operands cost over 20 steps
        loss_bound, drift_bound = (4e-7, 1e-6) if precision == "fp32" else (9e-4, 0.15)
        parity_note(f"g4b_loss_gap_{precision}", gap, loss_bound)
        parity_note(f"g4b_param_drift_{precision}", drift, drift_bound)
        assert gap < loss_bound, gap
        assert drift < drift_bound, drift
        assert int(m.step_count.item()) == int(g["step_count"])

    def test_bf16_step_tracks_amp_oracle(self, device, tmp_path):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer
        D, H, K, B = 128, 1024, 16, 256
        m, st = build(D, H, K, 21, False, 0.05, 100, "cpu", None)
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, warmup_steps=0, use_amp=True, num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        assert tr.use_amp
        x = synth.activations(B, D, seed=21, stream=6, bf16=False)
        met = tr.train_step(torch.from_numpy(x))
        r = O.train_step(st, x, 1e-3, "amp", max_norm=1.0)
        assert abs(met.loss - r["loss"]) / r["loss"] < 1e-5
        assert abs(met.grad_norm - r["grad_norm"]) / r["grad_norm"] < 1e-3
        sd = {k: cpu(v) for k, v in m.state_dict().items()}
        # first AdamW step moves each weight by ~lr*sign(g): compare updates, not values
        assert np.abs(sd["encoder.weight"] - st.W_e).max() < 2.5e-3
        assert np.mean(np.abs(sd["encoder.weight"] - st.W_e) < 1e-6) > 0.99


class TestDeadFeatures:
    def test_g6_four_alive_of_128(self, golden_dir, device):
        g = np.load(golden_dir / "g6_dead_tracking.npz")
        D, H, K = (int(v) for v in g["dims"])
        m, _ = build(D, H, K, 999, False, 0.0, 50, device, "fp32")
        m.train()
        x = torch.from_numpy(synth.activations(1, D, seed=999, stream=3, bf16=False)).to(device)
        assert m.get_dead_feature_ratio() == 0.0
        for _ in range(60):
            m(x)
        assert int(m.step_count.item()) == 60
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), g["last_activated"])
        assert int((~m.get_dead_features()).sum().item()) == int(g["alive"]) == 4
        assert abs(m.get_dead_feature_ratio() - float(g["ratio60"])) < 1e-7

    @pytest.mark.parametrize("tag,train_mode,num", [("train_all", True, None), ("eval_cap", False, 10),
                                                    ("train_many", True, None)])
    def test_g7_resample(self, golden_dir, device, tag, train_mode, num):
        g = np.load(golden_dir / "g7_resample.npz")
        D, H, K, B = (int(v) for v in g["dims"])
        m, _ = build(D, H, K, 5, False, 0.05, 20, device, "fp32")
        m.train(train_mode)
        with torch.no_grad():
            m.step_count.fill_(100)
            la = torch.full((H,), 95, dtype=torch.long)
            la[torch.from_numpy(g[f"{tag}.dead_idx"])] = 3
            m.feature_last_activated.copy_(la)
        x = torch.from_numpy(synth.activations(B, D, seed=5, stream=4, bf16=False)).to(device)
        ret = m.resample_dead_features(x, num)
        assert ret == int(g[f"{tag}.ret"])
        sd = {k: cpu(v) if v.is_floating_point() else v.cpu().numpy() for k, v in m.state_dict().items()}
        assert int(sd["step_count"]) == int(g[f"{tag}.step_count"])
        assert np.array_equal(sd["feature_last_activated"], g[f"{tag}.last_activated"])
        assert rel(sd["encoder.weight"], g[f"{tag}.W_e"]) < 1e-6
        assert rel(sd["decoder.weight"], g[f"{tag}.W_d"]) < 1e-6
        assert np.array_equal(sd["encoder.bias"] == 0, g[f"{tag}.b_e"] == 0)


class TestReLUSAE:
    """ReLU + L1 SAE (reference model.py:260-322) through wsae_relu_forward/backward."""

    def _build(self, D, H, seed, bf16, device, precision, weight=0.01):
        from whisper_sae.sae.model import ReLUSAE
        w = synth.sae_weights(D, H, seed=seed, bf16=bf16)
        m = ReLUSAE(D, H, sparsity_weight=weight, precision=precision)
        sd = m.state_dict()
        for key in ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias"):
            sd[key] = torch.from_numpy(w[key])
        m.load_state_dict(sd)
        return m.to(device), w

    def test_g8_forward_backward_fp32(self, golden_dir, device):
        g = np.load(golden_dir / "g8_relu.npz")
        D, H, B = (int(v) for v in g["dims"])
        m, _ = self._build(D, H, 11, False, device, "fp32")
        x = synth.activations(B, D, seed=11, stream=5, bf16=False)
        out = m(torch.from_numpy(x).to(device))
        out.loss.backward()
        assert abs(float(out.loss.detach()) - float(g["loss"])) / float(g["loss"]) < 1e-5
        assert abs(float(out.reconstruction_loss) - float(g["mse"])) / float(g["mse"]) < 1e-5
        assert abs(float(out.sparsity_loss) - float(g["l1"])) / float(g["l1"]) < 1e-5
        assert float(out.l0) == float(g["l0"])
        assert int((out.hidden > 0).sum().item()) == int(g["hidden_nnz"])
        assert rel(cpu(out.reconstructed), g["recon"]) < 1e-5
        for p, key in ((m.encoder.weight, "dW_e"), (m.encoder.bias, "db_e"), (m.decoder.weight, "dW_d"),
                       (m.decoder.bias, "db_d")):
            assert rel(cpu(p.grad), g[key]) < 2e-5, key

    # (the last three: whole 128-row groups, D % 128 == 0, H % 256 == 0 -> the bf16 row-major-GEMM flow of wsae_gemm256x.hip:
    # relu / dpre in the GEMM epilogues, no transposed copies; one of them with a partly filled last 256-row tile)
    @pytest.mark.parametrize("precision,tol,D,H,B", [("fp32", 2e-5, 96, 352, 200), ("bf16", 2e-2, 96, 352, 200),
                                                     ("fp32", 2e-5, 384, 3072, 1000), ("bf16", 2e-2, 384, 3072, 1000),
                                                     ("bf16", 2e-2, 384, 3072, 1024), ("bf16", 2e-2, 256, 1024, 640),
                                                     ("bf16", 2e-2, 128, 512, 256)])
    def test_ragged_shapes_against_the_oracle(self, device, precision, tol, D, H, B):
        """Batches and widths that are not multiples of the 64/128/256 tiles; the larger shape takes the persistent
        LDS-DMA GEMM (split-K for the two weight-gradient contractions), the smaller one the simple 128 x 128 kernel."""
        m, w = self._build(D, H, 5, True, device, precision, weight=0.05)
        x = synth.activations(B, D, seed=5, stream=2, bf16=True)
        out = m(torch.from_numpy(x).to(device))
        out.loss.backward()
        args = (w["encoder.weight"], w["encoder.bias"], w["decoder.weight"], w["decoder.bias"], x)
        f = O.relu_forward(*args, sparsity_weight=0.05)
        b = O.relu_backward(*args, f, sparsity_weight=0.05)
        ftol = 1e-5 if precision == "fp32" else 5e-3
        assert abs(float(out.loss.detach()) - float(f["loss"])) / float(f["loss"]) < ftol
        assert abs(float(out.sparsity_loss) - float(f["sparsity_loss"])) / float(f["sparsity_loss"]) < ftol
        assert rel(cpu(out.reconstructed), f["reconstructed"]) < ftol
        # a pre-activation within summation-order noise of zero may sit on either side of the relu in the two
        # computations; such a feature's dW_e / db_e row legitimately differs, so those rows are left out
        sure = (np.abs(f["pre"]) > 1e-6 * np.abs(f["pre"]).max()).all(axis=0)
        assert sure.mean() > 0.99
        for p, key in ((m.encoder.weight, "W_e"), (m.encoder.bias, "b_e"), (m.decoder.weight, "W_d"),
                       (m.decoder.bias, "b_d")):
            got, want = cpu(p.grad), b[key]
            if key in ("W_e", "b_e"):
                got, want = got[sure], want[sure]
            assert rel(got, want) < tol, key

    def test_trainer_step_matches_autograd_plus_adamw(self, device, tmp_path):
        """SAETrainer drives the ReLU module (the reference's trainer crashes on it): one fused step equals the
        oracle's gradients -> clip -> AdamW -> decoder renorm on the same weights."""
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer
        D, H, B = 64, 256, 32
        m, w = self._build(D, H, 11, False, device, "fp32")
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.0, epochs=1, warmup_steps=0,
                             gradient_clip=1.0, use_amp=False, num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        x = synth.activations(B, D, seed=11, stream=5, bf16=False)
        met = tr.train_step(torch.from_numpy(x).to(device))
        args = (w["encoder.weight"], w["encoder.bias"], w["decoder.weight"], w["decoder.bias"], x)
        f = O.relu_forward(*args, sparsity_weight=0.01)
        gr = O.relu_backward(*args, f, sparsity_weight=0.01)
        assert abs(met.loss - float(f["loss"])) / float(f["loss"]) < 1e-5
        assert abs(met.sparsity_loss - float(f["sparsity_loss"])) / float(f["sparsity_loss"]) < 1e-5
        assert abs(met.reconstruction_loss - float(f["reconstruction_loss"])) / float(f["reconstruction_loss"]) < 1e-4
        names = {"W_e": "encoder.weight", "b_e": "encoder.bias", "W_d": "decoder.weight", "b_d": "decoder.bias"}
        norm = np.sqrt(sum(float((gr[k].astype(np.float64) ** 2).sum()) for k in names))
        coef = min(1.0, 1.0 / (norm + 1e-6))
        sd = {k: cpu(v) for k, v in m.state_dict().items()}
        for k, name in names.items():
            gc = gr[k].astype(np.float64) * coef
            m1 = 0.1 * gc
            v1 = 0.001 * gc * gc
            upd = (m1 / 0.1) / (np.sqrt(v1 / 0.001) + 1e-8)
            want = w[name].astype(np.float64) - 1e-3 * upd
            if k == "W_d":
                want = want / np.maximum(np.linalg.norm(want, axis=0, keepdims=True), 1e-12)
            assert np.abs(sd[name] - want).max() < 2e-6, name
        assert float(m._engine.view("b_pre").abs().max().item()) == 0.0  # the unused pre-bias slot stays exactly zero


class TestDeterminism:
    def test_two_runs_of_three_steps_are_bit_identical(self, device, tmp_path):
        """No float atomics on the step path: every cross-block sum has a fixed order, so the same inputs
        give the same bits (bf16 mode, cfg-2 dimensions)."""
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer
        D, H, K, B = 384, 3072, 32, 1024
        packs = []
        for run in range(2):
            m, _ = build(D, H, K, 11, True, 0.05, 1000, device, "bf16")
            cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.0, epochs=1, warmup_steps=0,
                                 gradient_clip=1.0, use_amp=True, num_workers=0)
            tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path / f"r{run}")
            for step in range(3):
                x = torch.from_numpy(synth.activations(B, D, seed=11, stream=20 + step, bf16=True)).to(device)
                tr.train_step(x)
            torch.cuda.synchronize()
            packs.append(m._engine.pack.clone())
        assert torch.equal(packs[0], packs[1])


class TestDdpClock:
    """The one-collective dead-feature clock (include/wsae.h, wsae_ctx_set_fired) on a single GPU: the trainer
    is told it runs data-parallel and the 'all-reduce' is replaced by a function that adds what a second
    rank would have contributed to the fired indicators."""

    def test_fired_indicators_merge_like_all_reduce_max(self, device, tmp_path, monkeypatch):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae import training as T
        D, H, K, B = 64, 256, 8, 32
        m, st = build(D, H, K, 3, False, 0.05, 5, device, "fp32")
        cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.0, epochs=1, warmup_steps=0,
                             gradient_clip=1.0, use_amp=False, num_workers=0)
        tr = T.SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        x_all = synth.activations(2 * B, D, seed=3, stream=8, bf16=False)
        mine, theirs = x_all[:B], x_all[B:]
        # what the other rank's decode launch would have stamped in this step
        other = O.forward(st, theirs, "fp32", training=False)
        fired_other = torch.from_numpy(((other["hidden"] > 0).any(axis=0)).astype(np.float32)).to(device)
        seen = {}

        class FakeExchange:
            """Stands in for whisper_sae.distributed.WireExchange: the views the trainer hands over are 'summed' with what a
            second rank holding the same gradients and the other batch's fired indicators would have contributed."""

            def __init__(self):
                self.views = []

            def start(self, view):
                self.views.append(view)

            run = start  # (the in-stream collective of the single-launch backward: "summed" in finish() as well)

            def finish(self):
                torch.cuda.synchronize()
                from whisper_sae.distributed import WIRE_METRIC_SLOTS
                for v in self.views:
                    P = v.numel() - H - WIRE_METRIC_SLOTS  # ONE view = the whole wire [gradients | fired | metric digits]
                    seen["local"] = v[P:P + H].clone()
                    v[P:P + H] += fired_other  # SUM over the two ranks
                    v[:P] *= 2.0               # (both ranks hold this rank's gradients: sum = 2x, scale 1/2 below)
                    v[P + H:] *= 2.0           # ... and this rank's metric digits
                return 0.5

        monkeypatch.setattr(T, "world", lambda: (None, 2))
        monkeypatch.setattr(T, "WireExchange", FakeExchange)
        tr.train_step(torch.from_numpy(mine).to(device))
        torch.cuda.synchronize()
        fwd = O.forward(st, mine, "fp32", training=True)      # this rank's own stamps (step_count -> 1)
        local = (fwd["hidden"] > 0).any(axis=0)
        assert np.array_equal(seen["local"].cpu().numpy() > 0, local)  # decode wrote exactly its own features
        want = np.where(local | (fired_other.cpu().numpy() > 0), 1, 0).astype(np.int64)  # all_reduce(MAX) of the clocks
        assert np.array_equal(m.feature_last_activated.cpu().numpy(), want)
        assert float(tr.optimizer.fired.abs().sum().item()) == 0.0  # cleared for the next step
        assert int(m.step_count.item()) == 1


class TestTopKKernel:
    """The TopK kernel against the oracle's selection rule, including its exact (bisection) path."""

    def _run(self, device, pre_np, K):
        from whisper_sae.sae.model import TopKSAE
        B, H = pre_np.shape
        D = 32
        m = TopKSAE(D, H, k=K, precision="fp32").to(device)
        with torch.no_grad():  # identity-like encoder: pre == bias + 0, so feed pre through the bias of each row
            m.encoder.weight.zero_()
            m.b_pre.zero_()
        outs_v, outs_i = [], []
        for b in range(B):
            with torch.no_grad():
                m.encoder.bias.copy_(torch.from_numpy(pre_np[b]).to(device))
            v, i = m.encode_compact(torch.zeros(1, D, device=device))
            outs_v.append(cpu(v)[0])
            outs_i.append(i.cpu().numpy()[0])
        return np.stack(outs_v), np.stack(outs_i)

    @pytest.mark.parametrize("H,K", [(3072, 32), (256, 8), (128, 4), (12288, 64), (64, 64), (4096, 100)])
    def test_random_rows(self, device, H, K):
        pre = synth.normal((6, H), 77, H)
        v, i = self._run(device, pre, K)
        ov, oi = O.topk_select(pre, K)
        assert np.array_equal(i, oi) and np.array_equal(v, ov)

    def test_ties_lowest_index_first_and_fallback(self, device):
        H, K = 3072, 32
        rows = np.stack([
            np.zeros(H, np.float32),                                  # all equal: exact path
            np.repeat(np.arange(H // 512, dtype=np.float32), 512),   # 6 plateaus of 512 equal values
            np.where(np.arange(H) % 32 == 5, 1.0, -1.0).astype(np.float32),  # winners on one lane class
            -np.abs(synth.normal((H,), 5, 1)),                        # all negative
        ])
        v, i = self._run(device, rows, K)
        ov, oi = O.topk_select(rows, K)
        assert np.array_equal(i, oi) and np.array_equal(v, ov)


class TestTopKStrips:
    """The strip-guided TopK kernel (batches >= 2048 with H % 256 == 0: the persistent GEMM leaves the maxima of
    every 16-column strip, the TopK kernel reads only strips that can hold a winner) on rows built to stress it.
    Every batch row gets the same pre-activations (zero encoder weights, the row lives in the bias)."""

    def _run(self, device, row, K, precision):
        from whisper_sae.sae.model import TopKSAE
        H, D, B = row.shape[0], 64, 2048
        m = TopKSAE(D, H, k=K, precision=precision).to(device)
        with torch.no_grad():
            m.encoder.weight.zero_()
            m.b_pre.zero_()
            m.encoder.bias.copy_(torch.from_numpy(row).to(device))
        x = torch.from_numpy(synth.activations(B, D, seed=3, stream=1, bf16=True)).to(device)
        v, i = m.encode_compact(x)
        fb = int(m._engine.stats[6].item())
        v, i = cpu(v), i.cpu().numpy()
        assert (v == v[0]).all() and (i == i[0]).all()  # identical rows, identical answers
        return v[0], i[0], fb

    @pytest.mark.parametrize("precision,H,K", [("fp32", 3072, 32), ("bf16", 3072, 32), ("bf16", 12288, 64),
                                               ("fp32", 8192, 48)])
    def test_stress_rows(self, device, precision, H, K):
        rng = synth.normal((H,), 9, 3)
        rows = {
            "random": rng,
            "all_equal": np.full(H, 0.25, np.float32),                                  # every strip qualifies: exact path
            "plateaus": np.repeat(np.arange(H // 512, dtype=np.float32), 512),           # 512 ties at the top
            "one_strip": np.where(np.arange(H) // 16 == 77, 5.0 + np.arange(H) % 16, rng).astype(np.float32),
            "winners_spread": np.where(np.arange(H) % (H // K) == 7, 9.0, rng).astype(np.float32),  # K winners, K strips
            "negative": -np.abs(rng) - 1.0,
        }
        for name, row in rows.items():
            v, i, fb = self._run(device, row.astype(np.float32), K, precision)
            ov, oi = O.topk_select(row[None].astype(np.float32), K)
            assert np.array_equal(i, oi[0]) and np.array_equal(v, ov[0]), name
            if name in ("all_equal", "plateaus"):
                assert fb >= 2048, name   # these rows cannot be settled by the strip filter


class TestRing:
    def test_synthetic_fill_matches_oracle_generator(self, device):
        from whisper_sae.data import ActivationRing
        ring = ActivationRing(4096, 384, device=device, dtype=torch.bfloat16)
        ring.fill_synthetic(4096, seed=42)
        torch.cuda.synchronize()
        want = synth.activations(4096, 384, seed=42, stream=0, bf16=True)
        assert np.array_equal(ring.data.float().cpu().numpy(), want)
        ring32 = ActivationRing(512, 64, device=device, dtype=torch.float32)
        ring32.fill_synthetic(512, seed=7)
        assert np.array_equal(ring32.data.cpu().numpy(), synth.activations(512, 64, seed=7, stream=0, bf16=False))

    def test_sample_is_a_permutation_and_epochs_differ(self, device):
        from whisper_sae.data import ActivationRing, RingLoader
        ring = ActivationRing(1000, 32, device=device, dtype=torch.float32)
        ring.push(torch.from_numpy(synth.normal((1000, 32), 1, 0)))
        assert len(ring) == 1000
        p0 = ring.sample(1000, 42, 0, 0).cpu().numpy()
        p1 = ring.sample(1000, 42, 1, 0).cpu().numpy()
        assert sorted(p0.tolist()) == list(range(1000)) and sorted(p1.tolist()) == list(range(1000))
        assert not np.array_equal(p0, p1)
        assert np.array_equal(ring.sample(100, 42, 0, 300).cpu().numpy(), p0[300:400])
        loader = RingLoader(ring, 64, seed=42)
        assert len(loader) == 16  # ceil(1000/64), last batch partial
        sizes = [len(b) for b in loader]
        assert sizes == [64] * 15 + [40]
        # two ranks partition each global batch
        a = [b.rows.cpu().numpy() for b in RingLoader(ring, 50, seed=3, rank=0, world_size=2)]
        c = [b.rows.cpu().numpy() for b in RingLoader(ring, 50, seed=3, rank=1, world_size=2)]
        assert len(a) == len(c) == 10
        assert len(set(np.concatenate(a).tolist()) & set(np.concatenate(c).tolist())) == 0

    def test_ring_batch_trains_like_a_tensor_batch(self, device, tmp_path):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.data import ActivationRing
        from whisper_sae.sae.training import SAETrainer
        D, H, K, B = 64, 256, 8, 48
        x = synth.activations(400, D, seed=9, stream=0, bf16=True)
        ring = ActivationRing(400, D, device=device, dtype=torch.bfloat16)
        ring.fill_synthetic(400, seed=9)
        batch = ring.batch(B, 42, 0, 0)
        rows = batch.rows.cpu().numpy()
        losses = []
        for feed in ("ring", "tensor"):
            m, _ = build(D, H, K, 3, False, 0.05, 100, "cpu", None)
            tr = SAETrainer(m, TrainingConfig(batch_size=B, use_amp=True, warmup_steps=0, num_workers=0),
                            device=device, run_dir=tmp_path)
            met = tr.train_step(batch if feed == "ring" else torch.from_numpy(x[rows]))
            losses.append(met.loss)
            w = cpu(m.encoder.weight)
            losses.append(float(np.abs(w).sum()))
        assert losses[0] == losses[2] and losses[1] == losses[3]


class TestScaleProperties:
    """Size-independent properties at the bench configuration (cfg2: 384 -> 3072, k = 32)."""

    def test_large_batch_properties(self, device, tmp_path):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.data import ActivationRing
        from whisper_sae.sae.model import TopKSAE
        from whisper_sae.sae.training import SAETrainer
        D, H, K, B = 384, 3072, 32, 4096
        torch.manual_seed(42)
        m = TopKSAE(D, H, k=K, precision="bf16").to(device)
        ring = ActivationRing(1 << 16, D, device=device, dtype=torch.bfloat16)
        ring.fill_synthetic(1 << 16, seed=42)
        batch = ring.batch(B, 42, 0, 0)
        x = ring.data[batch.rows.long()]
        # selected set == torch.topk of the kernel's own dense pre-activations (ref test :110-130)
        pre = m.pre_activation(x)
        vals, idx = m.encode_compact(x)
        tv, ti = torch.topk(pre, K, dim=-1)
        assert torch.equal(torch.sort(idx.long(), dim=1).values, torch.sort(ti, dim=1).values)
        assert torch.equal(vals, tv)
        hidden = m.encode(x)
        assert torch.all((hidden != 0).sum(1) <= K) and torch.all(hidden >= 0)
        # linearity of decode in the code
        r1, r2 = m.decode(hidden), m.decode(2 * hidden)
        bias = m.decoder.bias + m.b_pre
        assert torch.allclose(r2 - bias, 2 * (r1 - bias), rtol=1e-5, atol=1e-6)
        # a training step keeps decoder columns at unit norm and lowers the loss on the same batch
        tr = SAETrainer(m, TrainingConfig(batch_size=B, learning_rate=1e-3, warmup_steps=0, use_amp=True,
                                          num_workers=0), device=device, run_dir=tmp_path)
        first = tr.train_step(batch).loss
        for _ in range(20):
            last = tr.train_step(batch).loss
        assert last < first
        cn = m.decoder.weight.norm(dim=0)
        assert torch.allclose(cn, torch.ones_like(cn), atol=1e-5)
        assert tr.metrics_history == [] and tr.global_step == 21


class TestEndToEnd:
    """scripts/train.py + FeatureCache + RingLoader + SAETrainer.train (cfg-1 plumbing, on the GPU)."""

    def test_cli_trains_from_a_cache(self, device, tmp_path):
        import importlib.util
        import json
        from whisper_sae.config import DataConfig, ExperimentConfig, SAEConfig, TrainingConfig, WandbConfig, WhisperConfig
        from whisper_sae.data import FeatureCache
        cfg = ExperimentConfig(
            sae=SAEConfig(expansion_factor=8, k=32, dead_feature_threshold=1000),
            training=TrainingConfig(batch_size=64, learning_rate=1e-3, epochs=3, warmup_steps=100, use_amp=True,
                                    checkpoint_every=2, num_workers=0),
            data=DataConfig(cache_dir=tmp_path / "cache"), wandb=WandbConfig(enabled=False),
            encoder_layers=[0], decoder_layers=[], output_dir=tmp_path / "out", experiment_name="e2e")
        cfg.to_yaml(tmp_path / "cfg.yaml")
        feats = torch.from_numpy(synth.activations(3000, 384, seed=1, stream=0, bf16=False))
        FeatureCache(cfg.data.cache_dir / "features", WhisperConfig(), cfg.data).save(feats, "encoder", 0, num_samples=2)
        spec = importlib.util.spec_from_file_location("wsae_train_cli", str(GOLDEN_ROOT / "scripts" / "train.py"))
        cli = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cli)
        cli.main(["--config", str(tmp_path / "cfg.yaml"), "--layer", "encoder:0", "--no-wandb"])
        run = tmp_path / "out" / "e2e_encoder_layer0"
        assert (run / "sae_final.pt").exists() and (run / "final.pt").exists() and (run / "checkpoint_epoch2.pt").exists()
        rows = json.loads((run / "metrics.json").read_text())
        assert len(rows) == 3 * 47 and rows[-1]["step"] == 141  # ceil(3000/64) = 47 steps per epoch
        first = np.mean([r["loss"] for r in rows[:47]])
        last = np.mean([r["loss"] for r in rows[-47:]])
        assert last < first  # ref test_training.py:214-240
        assert all(r["l0"] <= 32.0 for r in rows)
        sd = torch.load(run / "sae_final.pt", weights_only=True)
        assert sd["decoder.weight"].shape == (384, 3072) and sd["feature_last_activated"].dtype == torch.int64
        assert int(sd["step_count"]) == 141
