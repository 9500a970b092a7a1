"""Selective strip stores of the encoder GEMM (include/wsae.h ``wsae_ctx_set_strip_predict``; wsae_topk.h).

The persistent encoder GEMM writes a 16-column strip of the pre-activation matrix only when the strip's maximum reaches a
threshold predicted from the previous batch; the TopK launch checks every row against the same threshold and recomputes
what a row is missing.  The claim under test: the TopK code - values AND indices, every row - does not depend on the
prediction.  Bit-exact comparisons, three regimes:

* feature off vs feature on with its own history (the normal case: no row recomputes anything);
* a huge assumed threshold: the GEMM stores nothing, every row recomputes all its candidate strips in the TopK launch with
  the single-row MFMA loop - this is the test that the recomputation reproduces the GEMM's arithmetic bit for bit;
* a threshold in the middle of the batch's row thresholds: some rows recompute, some do not; and rows with more candidate
  strips than the strip path takes (constant rows -> the exact full-row path) on top of strips that were not stored.

Plus the trainer: a 6-step trajectory with the feature off equals the one with it on, bit for bit (loss and parameters),
and a 30-step run with drifting data refills nothing or next to nothing.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu

D, H, K = 384, 3072, 32
NAN = float("nan")


def build(device, seed=5, k=K, h=H, d=D):
    from whisper_sae.sae.model import TopKSAE
    w = synth.sae_weights(d, h, seed=seed, bf16=False, b_pre_scale=0.1)
    m = TopKSAE(d, h, k=k, precision="bf16")
    sd = m.state_dict()
    for key in ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias", "b_pre"):
        sd[key] = torch.from_numpy(w[key])
    m.load_state_dict(sd)
    return m.to(device)


def handle_of(m, B):
    from whisper_sae.sae.model import _precision_code
    eng = m.bind()
    return eng, eng.prepare(_precision_code(m.precision), B, force=True)


def predict(m, B, on, assume=NAN):
    from whisper_sae import _native as N
    eng, handle = handle_of(m, B)
    N.check(eng.lib.wsae_ctx_set_strip_predict(handle, int(on), C.c_float(assume)), "wsae_ctx_set_strip_predict")


def stats(m, B):
    from whisper_sae import _native as N
    eng, handle = handle_of(m, B)
    rows, tmin, margin = C.c_int64(0), C.c_float(0.0), C.c_float(0.0)
    N.check(eng.lib.wsae_ctx_strip_stats(handle, C.addressof(rows), C.addressof(tmin), C.addressof(margin)), "wsae_ctx_strip_stats")
    return rows.value, tmin.value, margin.value


def code(m, x):
    v, i = m.encode_compact(x)
    torch.cuda.synchronize()
    return v.clone(), i.clone()


def batch(device, B, seed, scale=1.0):
    x = synth.activations(B, D, seed=seed, stream=0, bf16=True) * scale
    return torch.from_numpy(x).to(device=device, dtype=torch.bfloat16)


class TestCodeDoesNotDependOnThePrediction:
    @pytest.mark.parametrize("B", [2048, 16384])
    def test_off_on_forced(self, device, B):
        m = build(device)
        x0, x1 = batch(device, B, 11), batch(device, B, 12)
        predict(m, B, False)
        ref_v, ref_i = code(m, x1)
        assert stats(m, B)[0] == 0

        # on, with its own history from another batch of the same distribution
        predict(m, B, True)
        code(m, x0)                                   # first launch: stores everything, leaves its minimum
        r0, tmin, _ = stats(m, B)
        assert r0 == 0 and tmin > 0
        v, i = code(m, x1)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)
        r1, tmin1, _ = stats(m, B)                      # tmin1: the smallest row threshold of x1
        refilled = r1 - r0
        assert refilled <= B // 1000, f"{refilled} rows recomputed strips on i.i.d. batches"

        # nothing stored at all: every row rebuilds its candidate strips itself
        predict(m, B, True, 1e30)
        v, i = code(m, x1)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)
        assert stats(m, B)[0] - r1 >= B

        # a store threshold 30 % above the smallest row threshold of this very batch: part of the rows are below it
        before = stats(m, B)[0]
        predict(m, B, True, 1.3 * tmin1)
        v, i = code(m, x1)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)
        some = stats(m, B)[0] - before
        assert 0 < some < B, f"{some} of {B} rows recomputed: the threshold was meant to split the batch"

    def test_history_from_a_larger_scale(self, device):
        """A batch three times smaller in scale right after a normal one: every row's T is below the predicted threshold."""
        B = 4096
        m = build(device)
        small = batch(device, B, 21, scale=0.3)
        predict(m, B, False)
        ref_v, ref_i = code(m, small)
        predict(m, B, True)
        code(m, batch(device, B, 22))
        before = stats(m, B)[0]
        v, i = code(m, small)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)
        assert stats(m, B)[0] - before > 0
        # ... and the next batch of the small scale is predicted from the small one: nothing to recompute
        before = stats(m, B)[0]
        predict(m, B, False)
        ref2 = code(m, batch(device, B, 23, scale=0.3))
        predict(m, B, True)
        code(m, small)
        before = stats(m, B)[0]
        got2 = code(m, batch(device, B, 23, scale=0.3))
        assert torch.equal(got2[1], ref2[1]) and torch.equal(got2[0], ref2[0])
        assert stats(m, B)[0] - before <= B // 1000

    def test_rows_that_take_the_exact_path(self, device):
        """Constant rows (every pre-activation of a row within a few distinct values -> more candidate strips than the strip
        path takes, and ties) between ordinary rows, with nothing stored: the full-row path must see a complete row."""
        B = 2048
        m = build(device)
        with torch.no_grad():
            m.encoder.bias.zero_()
            m.b_pre.zero_()
        x = batch(device, B, 31)
        x[100:164] = 0.0                   # pre = 0 everywhere: every strip is a candidate, every element ties
        x[500] = x[501]                    # duplicate rows
        predict(m, B, False)
        ref_v, ref_i = code(m, x)
        for assume in (1e30, 0.5, NAN):
            predict(m, B, True, assume)
            v, i = code(m, x)
            assert torch.equal(i, ref_i) and torch.equal(v, ref_v), f"assumed minimum {assume}"

    def test_fp32_input_goes_through_the_staged_copy(self, device):
        """fp32 activations are staged to bf16 first: the recompute path must read that staged copy, not the caller's tensor."""
        B = 2048
        m = build(device)
        x = batch(device, B, 51).float()
        predict(m, B, False)
        ref_v, ref_i = code(m, x)
        predict(m, B, True, 1e30)
        v, i = code(m, x)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)
        assert stats(m, B)[0] >= B

    def test_ring_rows_through_the_row_list(self, device, tmp_path):
        """The trainer's ring batches reach the GEMM as a row list into the ring: one forced-recompute step equals the plain one."""
        from whisper_sae.config import TrainingConfig
        from whisper_sae.data import ActivationRing
        from whisper_sae.sae.training import SAETrainer
        B = 4096
        out = []
        for assume in (None, 1e30):
            m = build(device, seed=9)
            cfg = TrainingConfig(batch_size=B, learning_rate=1e-4, weight_decay=0.0, epochs=1, warmup_steps=0, gradient_clip=1.0,
                                 use_amp=True, num_workers=0)
            tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
            ring = ActivationRing(1 << 15, D, device=device, dtype=torch.bfloat16)
            ring.fill_synthetic(1 << 15, seed=3)
            if assume is None:
                predict(m, B, False)
            else:
                predict(m, B, True, assume)
            met = tr.train_step(ring.batch(B, 42, 0, 0))
            torch.cuda.synchronize()
            out.append((float(met.loss), {k: v.detach().float().cpu().numpy().copy() for k, v in m.state_dict().items()
                                          if v.dtype.is_floating_point}, stats(m, B)[0]))
        assert out[0][2] == 0 and out[1][2] >= B
        assert out[0][0] == out[1][0]
        for k in out[0][1]:
            assert np.array_equal(out[0][1][k], out[1][1][k]), k

    @pytest.mark.parametrize("d,h,k,B", [(1280, 4096, 32, 2048),    # x row too long for the wave's LDS list: operands from memory
                                         (256, 16384, 64, 2048),    # 1024 strips per row (16 maxima per lane), two thresholds per lane
                                         (512, 2048, 16, 2304),     # 128 strips per row, the smallest the strip TopK takes
                                         (384, 3072, 32, 2048 + 272)])  # a ragged last tile (stores everything) beside full ones
    def test_other_shapes(self, device, d, h, k, B):
        from whisper_sae.sae.model import TopKSAE
        w = synth.sae_weights(d, h, seed=13, bf16=False, b_pre_scale=0.1)
        m = TopKSAE(d, h, k=k, precision="bf16")
        sd = m.state_dict()
        for key in ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias", "b_pre"):
            sd[key] = torch.from_numpy(w[key])
        m.load_state_dict(sd)
        m = m.to(device)
        xs = [torch.from_numpy(synth.activations(B, d, seed=60 + i, stream=0, bf16=True)).to(device=device, dtype=torch.bfloat16)
              for i in range(2)]
        predict(m, B, False)
        ref_v, ref_i = code(m, xs[1])
        predict(m, B, True)
        code(m, xs[0])
        _, tmin, _ = stats(m, B)
        v, i = code(m, xs[1])
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v), "own history"
        for assume in (1e30, 1.3 * tmin if tmin == tmin else 1.0):
            predict(m, B, True, assume)
            v, i = code(m, xs[1])
            assert torch.equal(i, ref_i) and torch.equal(v, ref_v), f"assumed threshold {assume}"

    def test_k64_two_maxima_per_lane(self, device):
        B = 2048
        m = build(device, k=64)
        x = batch(device, B, 41)
        predict(m, B, False)
        ref_v, ref_i = code(m, x)
        predict(m, B, True, 1e30)
        v, i = code(m, x)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)
        predict(m, B, True)
        code(m, batch(device, B, 42))      # history
        v, i = code(m, x)
        assert torch.equal(i, ref_i) and torch.equal(v, ref_v)


class TestTrainerTrajectory:
    def run(self, device, tmp_path, on, steps, B=4096, drift=False):
        from whisper_sae.config import TrainingConfig
        from whisper_sae.sae.training import SAETrainer
        m = build(device, seed=7)
        cfg = TrainingConfig(batch_size=B, learning_rate=3e-4, weight_decay=0.0, epochs=1, warmup_steps=2, gradient_clip=1.0,
                             use_amp=True, num_workers=0)
        tr = SAETrainer(m, cfg, device=device, run_dir=tmp_path)
        predict(m, B, on)
        losses, refills = [], []
        for s in range(steps):
            # drift: the scale of the data shrinks by 1 % per step AND alternates between two levels 30 % apart (a training
            # and a validation stream through one ctx): both are inside what the prediction absorbs
            scale = (1.0 - 0.01 * s) * (1.0 if s % 2 == 0 else 0.7) if drift else 1.0
            met = tr.train_step(batch(device, B, 100 + s, scale=scale))
            losses.append(float(met.loss))
            refills.append(stats(m, B)[0])
        torch.cuda.synchronize()
        params = {k: v.detach().float().cpu().numpy().copy() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
        return losses, params, refills

    def test_bit_identical_with_and_without(self, device, tmp_path):
        l_off, p_off, r_off = self.run(device, tmp_path, False, 6)
        l_on, p_on, r_on = self.run(device, tmp_path, True, 6)
        assert r_off[-1] == 0
        assert l_on == l_off
        for k in p_off:
            assert np.array_equal(p_on[k], p_off[k]), k

    def test_drifting_data_refills_next_to_nothing(self, device, tmp_path, parity_note):
        """After the first encounter of the second level (step 1: predicted from step 0's level, every row recomputes) the
        margin and the two-launch history keep up with the drift."""
        B, steps = 4096, 30
        _, _, refills = self.run(device, tmp_path, True, steps, B=B, drift=True)
        later = refills[-1] - refills[1]
        parity_note("strip_predict_refilled_rows_steps_2_to_29_of_4096", later, B * (steps - 2) // 1000)
        assert refills[1] > 0
        assert later <= B * (steps - 2) // 1000
