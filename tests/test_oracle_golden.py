"""Pin the CPU oracle (oracle/sae_oracle.py) against golden vectors produced by the real reference.

The goldens under tests/golden/ were written by tests/golden/make_golden.py, which imports
omarkhursheed/whisper-sae in the build container and records its outputs on inputs from
oracle/synth.py.  These tests regenerate the same inputs and require the numpy restatement to
reproduce the reference's outputs.  CPU only.
"""

from __future__ import annotations

import json

import numpy as np
import pytest

from oracle import sae_oracle as O
from oracle import synth

PARAM_KEYS = {"W_e": "encoder.weight", "b_e": "encoder.bias", "W_d": "decoder.weight",
              "b_d": "decoder.bias", "b_pre": "b_pre"}


def _state(D, H, k, seed, bf16, b_pre_scale, thr):
    w = synth.sae_weights(D, H, seed=seed, bf16=bf16, b_pre_scale=b_pre_scale)
    return O.SAEState.from_state_dict(w, k=k, dead_feature_threshold=thr)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


class TestSynth:
    def test_bf16_round_matches_torch(self):
        import torch
        x = synth.normal((4096,), 3, 0) * np.float32(37.0)
        ref = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
        assert np.array_equal(synth.bf16_round(x), ref)

    def test_streams_are_deterministic_and_distinct(self):
        a = synth.normal((1000,), 42, 1)
        assert np.array_equal(a, synth.normal((1000,), 42, 1))
        assert not np.array_equal(a, synth.normal((1000,), 42, 2))
        assert abs(float(a.mean())) < 0.15 and 0.85 < float(a.std()) < 1.15
        # first values are a fixed known answer (guards against silent generator changes)
        assert synth.counter_u64(2, 42, 0).tolist() == synth.counter_u64(4, 42, 0)[:2].tolist()


class TestForwardG1:
    @pytest.fixture(scope="class")
    def case(self, golden_dir):
        g = np.load(golden_dir / "g1_forward_cfg2.npz")
        D, H, K, B = g["dims"]
        st = _state(D, H, K, 42, True, 0.1, 1000)
        x = synth.activations(B, D, seed=42, stream=1, bf16=True)
        return g, st, x

    @pytest.mark.parametrize("mode", ["fp32", "amp"])
    def test_topk_sets_exact_and_values(self, case, mode):
        g, st, x = case
        out = O.forward(st.copy(), x, mode)
        assert g["min_margin"] > 1e-5
        # index SETS bit-exact (order inside the list may swap for interior near-equal values:
        # only the k/k+1 boundary carries the asserted margin)
        assert np.array_equal(np.sort(out["idx"].astype(np.int16), axis=1), np.sort(g["idx"], axis=1))
        assert rel(out["vals"], g["vals"]) < 1e-5  # both sorted descending

    @pytest.mark.parametrize("mode", ["fp32", "amp"])
    def test_recon_loss_l0(self, case, mode):
        g, st, x = case
        out = O.forward(st.copy(), x, mode)
        assert rel(out["reconstructed"], g["recon"]) < 1e-5
        assert abs(float(out["loss"]) - float(g["loss"])) / float(g["loss"]) < 1e-5
        assert float(out["l0"]) == float(g["l0"])

    def test_dead_tracking_after_one_forward(self, case):
        g, st, x = case
        s = st.copy()
        O.forward(s, x, "fp32", training=True)
        assert s.step_count == int(g["step_count"])
        assert np.array_equal(s.last_activated, g["last_activated"])


class TestGradsG2:
    @pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("amp", 2e-2)])
    def test_grads(self, golden_dir, mode, tol):
        g1 = np.load(golden_dir / "g1_forward_cfg2.npz")
        g2 = np.load(golden_dir / "g2_grads_cfg2.npz")
        D, H, K, B = g1["dims"]
        st = _state(D, H, K, 42, True, 0.1, 1000)
        x = synth.activations(B, D, seed=42, stream=1, bf16=True)
        fwd = O.forward(st, x, mode)
        gr = O.backward(st, x, fwd, mode)
        norms = [np.sqrt((gr[n].astype(np.float64) ** 2).sum()) for n in ("W_e", "b_e", "W_d", "b_d", "b_pre")]
        assert np.allclose(norms, g2["norms"], rtol=tol)
        for n in ("b_e", "b_d", "b_pre"):
            assert rel(gr[n], g2[n]) < tol, n
        assert rel(gr["W_e"].reshape(-1)[g2["pos_e"]], g2["W_e_samples"]) < tol * 5
        assert rel(gr["W_d"].reshape(-1)[g2["pos_d"]], g2["W_d_samples"]) < tol * 5


class TestTrainStepG3:
    def test_one_step(self, golden_dir):
        g1 = np.load(golden_dir / "g1_forward_cfg2.npz")
        g3 = np.load(golden_dir / "g3_train_step_cfg2.npz")
        D, H, K, B = g1["dims"]
        st = _state(D, H, K, 42, True, 0.1, 1000)
        x = synth.activations(B, D, seed=42, stream=1, bf16=True)
        lr0 = O.lr_at(0, 1e-4, 100, 35157)
        assert abs(lr0 - float(g3["lr0"])) < 1e-18 + 1e-12 * lr0
        r = O.train_step(st, x, lr0, "fp32", max_norm=1.0)
        assert abs(r["loss"] - float(g3["loss"])) / float(g3["loss"]) < 1e-5
        assert r["l0"] == float(g3["l0"])
        assert abs(r["grad_norm"] - float(g3["grad_total_norm"])) / float(g3["grad_total_norm"]) < 1e-5
        assert r["dead_feature_ratio"] == float(g3["dead_ratio"])
        # parameters after clip + AdamW + renorm.  AdamW's first step moves every weight by
        # ~lr * sign(g); tolerance is on the update, not just the (dominant) old value.
        for name, key in (("b_e", "b_e"), ("b_d", "b_d"), ("b_pre", "b_pre")):
            assert np.abs(getattr(st, name) - g3[key]).max() < 2e-7, name
        assert np.abs(st.W_e.reshape(-1)[g3["pos_e"]] - g3["W_e_samples"]).max() < 2e-7
        assert np.abs(st.W_d.reshape(-1)[g3["pos_d"]] - g3["W_d_samples"]).max() < 2e-6
        cn = np.linalg.norm(st.W_d.astype(np.float64), axis=0)
        assert abs(cn.min() - 1) < 1e-5 and abs(cn.max() - 1) < 1e-5
        assert abs(O.lr_at(1, 1e-4, 100, 35157) - float(g3["lr_after"])) < 1e-12 * 1e-4


class TestTrajectoryG4:
    def test_twenty_steps(self, golden_dir):
        g = np.load(golden_dir / "g4_trajectory_small.npz")
        D, H, K, B, STEPS = g["dims"]
        st = _state(D, H, K, 7, False, 0.05, 5)
        xs = synth.activations(B * STEPS, D, seed=7, stream=2, bf16=False).reshape(STEPS, B, D)
        for s in range(STEPS):
            lr = O.lr_at(s, 1e-3, 5, STEPS)
            assert abs(lr - g["lrs"][s]) < 1e-9 * 1e-3, s
            r = O.train_step(st, xs[s], lr, "fp32", max_norm=1.0, weight_decay=0.01)
            assert abs(r["loss"] - g["losses"][s]) / g["losses"][s] < 2e-5, s
            assert r["dead_feature_ratio"] == g["dead"][s], s
        assert rel(st.W_e, g["W_e"]) < 1e-4
        assert rel(st.W_d, g["W_d"]) < 1e-4
        assert rel(st.b_e, g["b_e"]) < 1e-4
        assert rel(st.b_pre, g["b_pre"]) < 1e-4
        assert np.array_equal(st.last_activated, g["last_activated"])
        assert st.step_count == int(g["step_count"])
        assert rel(st.adam_m["W_e"], g["exp_avg_We"]) < 1e-4
        assert rel(st.adam_v["W_e"], g["exp_avg_sq_We"]) < 1e-4


class TestTrajectoryG4b:
    """the reference's 20-step loss / lr scalars at cfg2 dimensions (384 -> 3072, k = 32, B = 512),
    bf16-representable inputs and initial weights.  Measured here (numpy oracle against the torch reference, both fp32 with
    different summation orders): worst relative loss gap 2.0e-5, sampled final parameters within 2.9e-4 of the tensor's
    largest entry - Adam's first updates are lr * m / sqrt(v) ~ lr * sign(g), so an entry whose gradient is within rounding
    of zero moves by up to 2 lr in either direction; the one-step pins (G1 - G3) hold 1e-5 / 2e-7, a multi-step trajectory
    cannot.  Bounds = measured x 3.  The "amp" mode (what the bf16 kernels mirror) is held to the SAME reference numbers:
    its measured gap, 3.9e-4 in the loss, is the price of bf16 operands over 20 steps and is recorded, not hidden."""

    @pytest.fixture(scope="class")
    def run(self, golden_dir):
        g = np.load(golden_dir / "g4b_trajectory_cfg2.npz")
        D, H, K, B, STEPS = (int(v) for v in g["dims"])
        w = synth.sae_weights(D, H, seed=11, bf16=True, b_pre_scale=0.1)
        xs = synth.activations(B * STEPS, D, seed=11, stream=4, bf16=True).reshape(STEPS, B, D)
        out = {}
        for mode in ("fp32", "amp"):
            st = O.SAEState.from_state_dict(w, k=K)
            gaps = []
            for s in range(STEPS):
                lr = O.lr_at(s, 1e-3, 5, STEPS)
                assert abs(lr - g["lrs"][s]) < 1e-9 * 1e-3, s
                r = O.train_step(st, xs[s], lr, mode, max_norm=1.0, weight_decay=0.0)
                gaps.append(abs(r["loss"] - g["losses"][s]) / g["losses"][s])
                assert r["l0"] == g["l0"][s] and r["dead_feature_ratio"] == g["dead"][s], (mode, s)
            out[mode] = (max(gaps), st)
        return g, out

    def test_fp32_mode_tracks_the_reference(self, run):
        g, out = run
        gap, st = out["fp32"]
        assert gap < 6e-5, gap  # measured 2.0e-5
        for short, arr in (("W_e", st.W_e), ("W_d", st.W_d), ("b_e", st.b_e), ("b_d", st.b_d), ("b_pre", st.b_pre)):
            got = arr.reshape(-1)[g[f"pos_{short}"]]
            assert np.abs(got - g[f"val_{short}"]).max() < 9e-4 * np.abs(g[f"val_{short}"]).max(), short  # measured <= 2.9e-4
            assert abs(np.linalg.norm(arr.astype(np.float64)) - g[f"norm_{short}"]) < 6e-6 * g[f"norm_{short}"], short  # 1.8e-6
        assert st.step_count == int(g["step_count"])

    def test_amp_mode_gap_to_the_reference_is_bounded(self, run):
        g, out = run
        gap, st = out["amp"]
        assert gap < 1.2e-3, gap  # measured 3.9e-4: bf16 operands, 20 steps
        assert abs(np.linalg.norm(st.W_e.astype(np.float64)) - g["norm_W_e"]) < 5e-5 * g["norm_W_e"]  # measured 1.4e-5


class TestLRScheduleG5:
    def test_all_cases(self, golden_dir):
        cases = json.loads((golden_dir / "g5_lr_schedule.json").read_text())
        for name, c in cases.items():
            for s, v in enumerate(c["values"]):
                got = O.lr_at(s, c["lr"], c["warmup_cfg"], c["total"])
                assert abs(got - v) <= 1e-9 * c["lr"], (name, s, got, v)


class TestDeadTrackingG6:
    def test_four_alive_of_128(self, golden_dir):
        g = np.load(golden_dir / "g6_dead_tracking.npz")
        D, H, K = g["dims"]
        st = _state(D, H, K, 999, False, 0.0, 50)
        x = synth.activations(1, D, seed=999, stream=3, bf16=False)
        assert O.dead_ratio(st) == float(g["ratio0"]) == 0.0
        for _ in range(60):
            O.forward(st, x, "fp32", training=True)
        assert st.step_count == int(g["step_count"]) == 60
        assert np.array_equal(st.last_activated, g["last_activated"])
        assert int((~O.dead_mask(st)).sum()) == int(g["alive"]) == 4
        assert abs(O.dead_ratio(st) - float(g["ratio60"])) < 1e-7


class TestResampleG7:
    @pytest.mark.parametrize("tag,train_mode,num", [("train_all", True, None), ("eval_cap", False, 10),
                                                    ("train_many", True, None)])
    def test_resample(self, golden_dir, tag, train_mode, num):
        g = np.load(golden_dir / "g7_resample.npz")
        D, H, K, B = g["dims"]
        st = _state(D, H, K, 5, False, 0.05, 20)
        x = synth.activations(B, D, seed=5, stream=4, bf16=False)
        st.step_count = 100
        st.last_activated[:] = 95
        st.last_activated[g[f"{tag}.dead_idx"]] = 3
        r = O.resample_dead_features(st, x, num, "fp32", training=train_mode)
        assert r["returned"] == int(g[f"{tag}.ret"])
        assert st.step_count == int(g[f"{tag}.step_count"])
        assert np.array_equal(st.last_activated, g[f"{tag}.last_activated"])
        assert rel(st.W_e, g[f"{tag}.W_e"]) < 1e-6
        assert rel(st.W_d, g[f"{tag}.W_d"]) < 1e-6
        assert np.array_equal(st.b_e == 0, g[f"{tag}.b_e"] == 0)
        if tag == "train_many":  # capped count returned, only B rows rewritten (model.py:243-257)
            assert r["returned"] == 52 and len(r["rewritten"]) == B


class TestReLUG8:
    def test_relu_forward_backward(self, golden_dir):
        g = np.load(golden_dir / "g8_relu.npz")
        D, H, B = g["dims"]
        w = synth.sae_weights(D, H, seed=11, bf16=False)
        x = synth.activations(B, D, seed=11, stream=5, bf16=False)
        args = (w["encoder.weight"], w["encoder.bias"], w["decoder.weight"], w["decoder.bias"], x)
        f = O.relu_forward(*args, sparsity_weight=0.01)
        assert abs(float(f["loss"]) - float(g["loss"])) / float(g["loss"]) < 1e-5
        assert abs(float(f["reconstruction_loss"]) - float(g["mse"])) / float(g["mse"]) < 1e-5
        assert abs(float(f["sparsity_loss"]) - float(g["l1"])) / float(g["l1"]) < 1e-5
        assert float(f["l0"]) == float(g["l0"])
        assert rel(f["reconstructed"], g["recon"]) < 1e-5
        b = O.relu_backward(*args, f, sparsity_weight=0.01)
        for n, key in (("W_e", "dW_e"), ("b_e", "db_e"), ("W_d", "dW_d"), ("b_d", "db_d")):
            assert rel(b[n], g[key]) < 2e-5, n


def test_fp8_e4m3_rounding_equals_torch_float8():
    """The oracle's ``"fp8"`` mode (ReLU forward, BASELINE.json configs[4]) rounds like ``torch.float8_e4m3fn``."""
    import torch
    rng = np.random.default_rng(0)
    for scale in (100.0, 1.0, 0.01):
        v = np.clip(rng.normal(size=200000) * scale, -448, 448).astype(np.float32)
        want = torch.from_numpy(v).to(torch.float8_e4m3fn).float().numpy()
        assert np.array_equal(O.fp8_e4m3_round(v), want)
    ties = np.array([17.0, 19.0, 1.0625, 1.1875, 0.0009765625, 448.0, 463.9, -17.0], np.float32)
    assert np.array_equal(O.fp8_e4m3_round(ties), torch.from_numpy(ties).to(torch.float8_e4m3fn).float().numpy())
    q, s = O.fp8_quant_rows(np.array([[1.0, -3.0, 0.5], [0.0, 0.0, 0.0]], np.float32))
    assert q[0, 1] == -448.0 and s[0] == np.float32(3.0) / np.float32(448.0) and s[1] == 1.0 and (q[1] == 0).all()
