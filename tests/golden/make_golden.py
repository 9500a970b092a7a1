#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference in this container.

Run (build container only -- the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

The reference (omarkhursheed/whisper-sae @ /root/reference) ships no golden vectors, KATs or
``.pt`` fixtures for its SAE path: its tests are property tests on random data.
So the oracle is pinned with outputs of the reference itself, produced here.  Inputs and weights
come from ``oracle/synth.py`` (integer counter generator, version-proof) and are loaded into the
reference modules with ``load_state_dict``; only the reference's *outputs* are stored.  Everything
written is data (npz / json): no reference source text is copied anywhere.

Golden sets (SURVEY.md row C list):
  G1 forward @ cfg2 dims      G2 gradients            G3 one SAETrainer.train_step
  G4 20-step trajectory       G5 LR schedule          G6 dead-feature tracking
  G7 resample_dead_features   G8 ReLUSAE fwd/grads    G9 API bookkeeping (keys, batch forms)
  G10 seeded initialisation (torch.manual_seed(42) -> TopKSAE / ReLUSAE parameters)
  G11 FeatureCache interchange: a cache written by the reference's FeatureCache.save (N1)
  G12 transcoders: TopKTranscoder / SkipTranscoder forward, gradients, resample (N3)
  G13 per-feature top activations: the reference's TopKTracker over three updates (N4)
  G14 activation producer: extract_features_batch on a seeded random-init tiny Whisper (N2)
  G15 TopKCrossLayerCrosscoder: seeded init, forward, every gradient, clock, decoder helpers (N4 sibling)
  G16 CrossLayerCrosscoder (ReLU + decoder-norm-weighted L1): forward, every gradient, clock
  G17 extraction driver: extract_and_cache_features on the seeded tiny Whisper -> cache tensors + metadata (N2)
  G4b 20-step trajectory at cfg2 dimensions (384 -> 3072, k = 32): loss / lr / l0 scalars + sampled final parameters

``python tests/golden/make_golden.py g10 g11`` regenerates only the named sets.
"""

from __future__ import annotations

import json
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True  # the reference tree is read-only input: importing it must not leave __pycache__ there

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))

from oracle import synth  # noqa: E402

from whisper_sae.config import TrainingConfig  # noqa: E402  (reference)
from whisper_sae.sae.model import ReLUSAE, TopKSAE  # noqa: E402  (reference)
from whisper_sae.sae.training import SAETrainer  # noqa: E402  (reference)

torch.set_num_threads(4)


def load_weights(model, w):
    sd = model.state_dict()
    for k_, v in w.items():
        sd[k_] = torch.from_numpy(np.array(v))
    model.load_state_dict(sd)


def sample_positions(shape, n, seed):
    total = int(np.prod(shape))
    pos = (synth.counter_u64(n, seed, 99) % np.uint64(total)).astype(np.int64)
    return pos


def sd_numpy(model):
    return {k_: v.detach().cpu().numpy().copy() for k_, v in model.state_dict().items()}


def g1_g2_g3():
    D, H, K, B = 384, 3072, 32, 64
    w = synth.sae_weights(D, H, seed=42, bf16=True, b_pre_scale=0.1)
    x = synth.activations(B, D, seed=42, stream=1, bf16=True)
    torch.manual_seed(0)
    m = TopKSAE(D, H, k=K, dead_feature_threshold=1000)
    load_weights(m, w)
    m.train()
    xt = torch.from_numpy(x)
    # pre-activations for the margin check (same arithmetic as model.py:108-111)
    with torch.no_grad():
        pre = m.encoder(xt - m.b_pre).numpy()
    margin = synth.topk_margin(pre, K)
    assert margin.min() > 1e-5, f"fixture has a near-tie at the k/k+1 boundary: {margin.min()}"
    out = m(xt)
    vals, idx = torch.topk(torch.from_numpy(pre), K, dim=-1)
    hid = out.hidden.detach().numpy()
    # the reference's hidden must be exactly the scatter of relu(topk)
    chk = np.zeros_like(hid)
    np.put_along_axis(chk, idx.numpy(), np.maximum(vals.numpy(), 0), axis=1)
    assert np.array_equal(chk, hid)
    out.loss.backward()
    grads = {"W_e": m.encoder.weight.grad, "b_e": m.encoder.bias.grad, "W_d": m.decoder.weight.grad,
             "b_d": m.decoder.bias.grad, "b_pre": m.b_pre.grad}
    grads = {k_: v.detach().numpy().copy() for k_, v in grads.items()}
    pos_e = sample_positions((H, D), 1024, 7)
    pos_d = sample_positions((D, H), 1024, 8)
    np.savez_compressed(
        HERE / "g1_forward_cfg2.npz",
        dims=np.array([D, H, K, B]), seed=np.array([42]), b_pre_scale=np.array([0.1]),
        idx=idx.numpy().astype(np.int16), vals=vals.numpy(), recon=out.reconstructed.detach().numpy(),
        loss=np.float32(out.loss.item()), l0=np.float32(out.l0.item()), min_margin=np.float64(margin.min()),
        last_activated=m.feature_last_activated.numpy().copy(), step_count=np.int64(m.step_count.item()),
    )
    np.savez_compressed(
        HERE / "g2_grads_cfg2.npz",
        norms=np.array([np.sqrt((grads[n].astype(np.float64) ** 2).sum()) for n in ("W_e", "b_e", "W_d", "b_d", "b_pre")]),
        b_e=grads["b_e"], b_d=grads["b_d"], b_pre=grads["b_pre"],
        pos_e=pos_e, W_e_samples=grads["W_e"].reshape(-1)[pos_e],
        pos_d=pos_d, W_d_samples=grads["W_d"].reshape(-1)[pos_d],
    )

    # G3: one full SAETrainer.train_step from the same start (fresh model: dead counters at 0)
    m2 = TopKSAE(D, H, k=K, dead_feature_threshold=1000)
    load_weights(m2, w)
    cfg = TrainingConfig(batch_size=B, learning_rate=1e-4, weight_decay=0.0, epochs=3, warmup_steps=100,
                         gradient_clip=1.0, use_amp=True, checkpoint_every=2, seed=42, num_workers=0)
    with tempfile.TemporaryDirectory() as td:
        tr = SAETrainer(m2, cfg, device="cpu", run_dir=Path(td))
        tr.setup_scheduler(35157)
        lr0 = tr.optimizer.param_groups[0]["lr"]
        met = tr.train_step(torch.from_numpy(x))
        # also: the tuple / list batch forms give the same step (G9)
        met_t = tr.train_step((torch.from_numpy(x),))
        met_l = tr.train_step([torch.from_numpy(x)])
    # re-run a single step on a fresh copy for the post-step-1 parameter values
    m3 = TopKSAE(D, H, k=K, dead_feature_threshold=1000)
    load_weights(m3, w)
    with tempfile.TemporaryDirectory() as td:
        tr3 = SAETrainer(m3, cfg, device="cpu", run_dir=Path(td))
        tr3.setup_scheduler(35157)
        met3 = tr3.train_step(torch.from_numpy(x))
        ck = tr3.save_checkpoint("c.pt")
        ck_keys = sorted(torch.load(ck, weights_only=False).keys())
        opt_sd = tr3.optimizer.state_dict()
    sd = sd_numpy(m3)
    total_norm = float(np.sqrt(sum((grads[n].astype(np.float64) ** 2).sum() for n in grads)))
    np.savez_compressed(
        HERE / "g3_train_step_cfg2.npz",
        lr0=np.float64(lr0), loss=np.float64(met3.loss), l0=np.float64(met3.l0),
        dead_ratio=np.float64(met3.dead_feature_ratio), lr_after=np.float64(met3.learning_rate),
        step=np.int64(met3.step), grad_total_norm=np.float64(total_norm),
        b_e=sd["encoder.bias"], b_d=sd["decoder.bias"], b_pre=sd["b_pre"],
        pos_e=pos_e, W_e_samples=sd["encoder.weight"].reshape(-1)[pos_e],
        pos_d=pos_d, W_d_samples=sd["decoder.weight"].reshape(-1)[pos_d],
        W_e_norm=np.float64(np.sqrt((sd["encoder.weight"].astype(np.float64) ** 2).sum())),
        W_d_colnorm_minmax=np.array([np.linalg.norm(sd["decoder.weight"], axis=0).min(),
                                     np.linalg.norm(sd["decoder.weight"], axis=0).max()]),
        losses_3steps=np.array([met.loss, met_t.loss, met_l.loss]),
        lrs_3steps=np.array([met.learning_rate, met_t.learning_rate, met_l.learning_rate]),
    )
    api = {
        "state_dict_keys": list(m3.state_dict().keys()),
        "state_dict_shapes": {k_: list(v.shape) for k_, v in m3.state_dict().items()},
        "state_dict_dtypes": {k_: str(v.dtype) for k_, v in m3.state_dict().items()},
        "checkpoint_keys": ck_keys,
        "optimizer_param_group_keys": sorted(opt_sd["param_groups"][0].keys()),
        "optimizer_state_keys": sorted(opt_sd["state"][0].keys()),
        "metrics_fields": list(met3.__dataclass_fields__.keys()),
    }
    (HERE / "g9_api.json").write_text(json.dumps(api, indent=1))


def g4_trajectory():
    D, H, K, B, STEPS = 64, 256, 8, 16, 20
    w = synth.sae_weights(D, H, seed=7, bf16=False, b_pre_scale=0.05)
    xs = synth.activations(B * STEPS, D, seed=7, stream=2, bf16=False).reshape(STEPS, B, D)
    m = TopKSAE(D, H, k=K, dead_feature_threshold=5)
    load_weights(m, w)
    cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.01, epochs=1, warmup_steps=5,
                         gradient_clip=1.0, use_amp=False, num_workers=0)
    losses, lrs, dead, margins = [], [], [], []
    with tempfile.TemporaryDirectory() as td:
        tr = SAETrainer(m, cfg, device="cpu", run_dir=Path(td))
        tr.setup_scheduler(STEPS)
        for s in range(STEPS):
            with torch.no_grad():
                pre = m.encoder(torch.from_numpy(xs[s]) - m.b_pre).numpy()
            margins.append(synth.topk_margin(pre, K).min())
            lrs.append(tr.optimizer.param_groups[0]["lr"])
            met = tr.train_step(torch.from_numpy(xs[s]))
            losses.append(met.loss)
            dead.append(met.dead_feature_ratio)
        opt = tr.optimizer.state_dict()
    assert min(margins) > 1e-4, f"trajectory has a near-tie: {min(margins)}"
    sd = sd_numpy(m)
    np.savez_compressed(
        HERE / "g4_trajectory_small.npz",
        dims=np.array([D, H, K, B, STEPS]), losses=np.array(losses, dtype=np.float64),
        lrs=np.array(lrs, dtype=np.float64), dead=np.array(dead, dtype=np.float64),
        min_margin=np.float64(min(margins)),
        W_e=sd["encoder.weight"], b_e=sd["encoder.bias"], W_d=sd["decoder.weight"], b_d=sd["decoder.bias"],
        b_pre=sd["b_pre"], last_activated=sd["feature_last_activated"], step_count=sd["step_count"],
        exp_avg_We=opt["state"][1]["exp_avg"].numpy(), exp_avg_sq_We=opt["state"][1]["exp_avg_sq"].numpy(),  # parameters() order: b_pre, enc.W, enc.b, dec.W, dec.b
    )


def g4b_trajectory_cfg2():
    """SURVEY.md row C, "G4 ... loss/lr scalars at cfg2 dims": 20 reference steps at 384 -> 3072, k = 32 on
    bf16-representable inputs and initial weights (what both arithmetic modes of the build can be fed unchanged).  Stored:
    loss / lr / l0 / dead ratio per step, the smallest k / k+1 margin met, and of the final state the per-tensor norms plus a
    few hundred sampled entries (scalars at these dimensions, not the 19 MB of tensors)."""
    D, H, K, B, STEPS = 384, 3072, 32, 512, 20
    w = synth.sae_weights(D, H, seed=11, bf16=True, b_pre_scale=0.1)
    xs = synth.activations(B * STEPS, D, seed=11, stream=4, bf16=True).reshape(STEPS, B, D)
    m = TopKSAE(D, H, k=K)
    load_weights(m, w)
    cfg = TrainingConfig(batch_size=B, learning_rate=1e-3, weight_decay=0.0, epochs=1, warmup_steps=5,
                         gradient_clip=1.0, use_amp=False, num_workers=0)
    losses, lrs, l0s, dead, margins = [], [], [], [], []
    with tempfile.TemporaryDirectory() as td:
        tr = SAETrainer(m, cfg, device="cpu", run_dir=Path(td))
        tr.setup_scheduler(STEPS)
        for s in range(STEPS):
            with torch.no_grad():
                pre = m.encoder(torch.from_numpy(xs[s]) - m.b_pre).numpy()
            margins.append(synth.topk_margin(pre, K).min())
            lrs.append(tr.optimizer.param_groups[0]["lr"])
            met = tr.train_step(torch.from_numpy(xs[s]))
            losses.append(met.loss)
            l0s.append(met.l0)
            dead.append(met.dead_feature_ratio)
    sd = sd_numpy(m)
    out = {"dims": np.array([D, H, K, B, STEPS]), "losses": np.array(losses, dtype=np.float64),
           "lrs": np.array(lrs, dtype=np.float64), "l0": np.array(l0s, dtype=np.float64),
           "dead": np.array(dead, dtype=np.float64), "min_margin": np.float64(min(margins)),
           "margins": np.array(margins, dtype=np.float64),
           "step_count": sd["step_count"], "last_activated": sd["feature_last_activated"]}
    for key, short in (("encoder.weight", "W_e"), ("encoder.bias", "b_e"), ("decoder.weight", "W_d"),
                       ("decoder.bias", "b_d"), ("b_pre", "b_pre")):
        a = sd[key]
        out[f"norm_{short}"] = np.float64(np.linalg.norm(a.astype(np.float64)))
        pos = sample_positions(a.shape, min(256, a.size), seed=100 + len(short))
        out[f"pos_{short}"] = pos
        out[f"val_{short}"] = a.reshape(-1)[pos]
    np.savez_compressed(HERE / "g4b_trajectory_cfg2.npz", **out)


def g5_lr():
    out = {}
    for name, (total, warm, lr) in {"cfg1": (35157, 100, 1e-4), "short": (50, 1000, 1e-3),
                                    "nowarm": (40, 0, 3e-4), "tiny": (9, 100, 1e-4)}.items():
        m = TopKSAE(32, 64, k=4)
        cfg = TrainingConfig(learning_rate=lr, warmup_steps=warm, use_amp=False, num_workers=0)
        with tempfile.TemporaryDirectory() as td:
            tr = SAETrainer(m, cfg, device="cpu", run_dir=Path(td))
            tr.setup_scheduler(total)
            n = min(total, 400)
            vals = []
            for _ in range(n):
                vals.append(tr.optimizer.param_groups[0]["lr"])
                tr.optimizer.step()
                tr.scheduler.step()
        out[name] = {"total": total, "warmup_cfg": warm, "lr": lr, "values": vals}
    (HERE / "g5_lr_schedule.json").write_text(json.dumps(out))


def g6_dead():
    # the reference's own "4 alive of 128 after 60 steps" scenario (tests/test_sae_model.py:251-294)
    D, H, K = 32, 128, 4
    w = synth.sae_weights(D, H, seed=999, bf16=False)
    x = synth.activations(1, D, seed=999, stream=3, bf16=False)
    m = TopKSAE(D, H, k=K, dead_feature_threshold=50)
    load_weights(m, w)
    m.train()
    r0 = m.get_dead_feature_ratio()
    for _ in range(60):
        m(torch.from_numpy(x))
    np.savez_compressed(
        HERE / "g6_dead_tracking.npz", dims=np.array([D, H, K]), ratio0=np.float64(r0),
        ratio60=np.float64(m.get_dead_feature_ratio()), alive=np.int64((~m.get_dead_features()).sum().item()),
        last_activated=m.feature_last_activated.numpy().copy(), step_count=np.int64(m.step_count.item()),
    )


def g7_resample():
    D, H, K, B = 64, 256, 8, 24
    w = synth.sae_weights(D, H, seed=5, bf16=False, b_pre_scale=0.05)
    x = synth.activations(B, D, seed=5, stream=4, bf16=False)
    res = {}
    for tag, train_mode, num in (("train_all", True, None), ("eval_cap", False, 10), ("train_many", True, None)):
        m = TopKSAE(D, H, k=K, dead_feature_threshold=20)
        load_weights(m, w)
        m.train(train_mode)
        with torch.no_grad():
            m.step_count.fill_(100)
            la = torch.full((H,), 95, dtype=torch.long)
            n_dead = 52 if tag == "train_many" else 12  # 52 > B: "returns capped count, rewrites B" quirk
            dead_idx = (synth.counter_u64(400, 5, 50) % np.uint64(H)).astype(np.int64)
            dead_idx = np.unique(dead_idx)[:n_dead]
            la[torch.from_numpy(dead_idx)] = 3
            m.feature_last_activated.copy_(la)
        ret = m.resample_dead_features(torch.from_numpy(x), num)
        sd = sd_numpy(m)
        res[tag] = dict(ret=ret, dead_idx=dead_idx, W_e=sd["encoder.weight"], b_e=sd["encoder.bias"],
                        W_d=sd["decoder.weight"], last_activated=sd["feature_last_activated"],
                        step_count=sd["step_count"])
    flat = {}
    for tag, d in res.items():
        for k_, v in d.items():
            flat[f"{tag}.{k_}"] = np.asarray(v)
    np.savez_compressed(HERE / "g7_resample.npz", dims=np.array([D, H, K, B]), **flat)


def g8_relu():
    D, H, B = 64, 256, 32
    w = synth.sae_weights(D, H, seed=11, bf16=False)
    x = synth.activations(B, D, seed=11, stream=5, bf16=False)
    m = ReLUSAE(D, H, sparsity_weight=0.01)
    sd = m.state_dict()
    for k_ in ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias"):
        sd[k_] = torch.from_numpy(w[k_])
    m.load_state_dict(sd)
    out = m(torch.from_numpy(x))
    out.loss.backward()
    np.savez_compressed(
        HERE / "g8_relu.npz", dims=np.array([D, H, B]), loss=np.float32(out.loss.item()),
        mse=np.float32(out.reconstruction_loss.item()), l1=np.float32(out.sparsity_loss.item()),
        l0=np.float32(out.l0.item()), recon=out.reconstructed.detach().numpy(),
        hidden_nnz=np.int64((out.hidden > 0).sum().item()),
        dW_e=m.encoder.weight.grad.numpy(), db_e=m.encoder.bias.grad.numpy(),
        dW_d=m.decoder.weight.grad.numpy(), db_d=m.decoder.bias.grad.numpy(),
    )


def tensor_digest(a: np.ndarray) -> np.ndarray:
    """Order-sensitive 64-bit digest of a tensor's exact bit pattern: [sum of words, sum of (i+1)*word] mod 2^64."""
    u = np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).reshape(-1)
    with np.errstate(over="ignore"):
        w = (np.arange(u.size, dtype=np.uint64) + np.uint64(1)) * u
        return np.array([u.sum(dtype=np.uint64), w.sum(dtype=np.uint64)], dtype=np.uint64)


def g10_seeded_init():
    """scripts/train.py:84-90,:248 seeds torch and then builds the model: the drop-in module must draw the same
    parameters for the same seed (same construction and RNG order as model.py:38-89 / :266-286)."""
    out = {"torch_version": np.array(torch.__version__)}
    for tag, (D, H, K) in {"cfg2": (384, 3072, 32), "small": (64, 256, 8)}.items():
        torch.manual_seed(42)
        m = TopKSAE(D, H, k=K)
        for k_, v in sd_numpy(m).items():
            if v.dtype == np.float32:
                out[f"topk.{tag}.{k_}.digest"] = tensor_digest(v)
                out[f"topk.{tag}.{k_}.head"] = v.reshape(-1)[:64].copy()
        nxt = torch.rand(4).numpy()  # the RNG position after construction (the module drew exactly as many numbers)
        out[f"topk.{tag}.next_rand"] = nxt
        torch.manual_seed(42)
        r = ReLUSAE(D, H, sparsity_weight=0.01)
        for k_, v in sd_numpy(r).items():
            out[f"relu.{tag}.{k_}.digest"] = tensor_digest(v)
            out[f"relu.{tag}.{k_}.head"] = v.reshape(-1)[:64].copy()
        out[f"relu.{tag}.next_rand"] = torch.rand(4).numpy()
    np.savez_compressed(HERE / "g10_seeded_init.npz", **out)


def g11_cache_interchange():
    """A small activation cache written by the REFERENCE's FeatureCache.save (feature_cache.py:136-167): the
    ``.pt`` tensor file and its ``_meta.json`` sidecar are data in the reference's on-disk format."""
    from whisper_sae.config import DataConfig, WhisperConfig  # reference
    from whisper_sae.data.feature_cache import FeatureCache  # reference
    target = HERE / "g11_cache"
    target.mkdir(exist_ok=True)
    feats = torch.from_numpy(synth.activations(96, 384, seed=77, stream=0, bf16=False))
    fc = FeatureCache(target, WhisperConfig(), DataConfig(cache_dir=Path("cache")))
    fc.save(feats, "encoder", 0, num_samples=2)
    meta_path = fc._get_metadata_path("encoder", 0)
    meta = json.loads(meta_path.read_text())
    meta["created_at"] = "2026-01-01T00:00:00"  # pinned so that regenerating the fixture is byte-stable
    meta_path.write_text(json.dumps(meta, indent=2))


def g12_transcoders():
    """TopKTranscoder / SkipTranscoder (sae/transcoder.py): forward, every gradient, and one resample call."""
    from whisper_sae.sae.transcoder import SkipTranscoder, TopKTranscoder  # reference
    out = {}
    for tag, (Din, Dout, H, K, B) in {"eq": (64, 64, 256, 8, 48), "narrow": (64, 32, 128, 8, 40)}.items():
        w = synth.sae_weights(Din, H, seed=23, bf16=False)
        wd = synth.normal((Dout, H), 23, 33).astype(np.float64)
        wd = (wd / np.sqrt((wd * wd).sum(axis=0, keepdims=True)) * 0.1).astype(np.float32)
        bd = synth.uniform((Dout,), 23, 34, -0.05, 0.05)
        x = synth.activations(B, Din, seed=23, stream=6, bf16=False)
        tgt = synth.activations(B, Dout, seed=23, stream=7, bf16=False)
        m = TopKTranscoder(Din, Dout, H, k=K, dead_feature_threshold=20)
        sd = m.state_dict()
        sd["encoder.weight"], sd["encoder.bias"] = torch.from_numpy(w["encoder.weight"]), torch.from_numpy(w["encoder.bias"])
        sd["decoder.weight"], sd["decoder.bias"] = torch.from_numpy(wd), torch.from_numpy(bd)
        m.load_state_dict(sd)
        m.train()
        xt = torch.from_numpy(x).requires_grad_(True)
        o = m(xt, torch.from_numpy(tgt))
        o.loss.backward()
        out.update({f"{tag}.dims": np.array([Din, Dout, H, K, B]), f"{tag}.W_d": wd, f"{tag}.b_d": bd,
                    f"{tag}.pred": o.predicted.detach().numpy(), f"{tag}.loss": np.float32(o.loss.item()),
                    f"{tag}.l0": np.float32(o.l0.item()),
                    f"{tag}.idx": np.sort(torch.topk(m.encoder(torch.from_numpy(x)), K, dim=-1).indices.numpy(), axis=1).astype(np.int16),
                    f"{tag}.dW_e": m.encoder.weight.grad.numpy(), f"{tag}.db_e": m.encoder.bias.grad.numpy(),
                    f"{tag}.dW_d": m.decoder.weight.grad.numpy(), f"{tag}.db_d": m.decoder.bias.grad.numpy(),
                    f"{tag}.dx": xt.grad.numpy(), f"{tag}.step_count": np.int64(m.step_count.item()),
                    f"{tag}.last_activated": m.feature_last_activated.numpy().copy()})
        if tag == "eq":
            # resample on a crafted dead state (train mode: the forward inside bumps the clock)
            m2 = TopKTranscoder(Din, Dout, H, k=K, dead_feature_threshold=20)
            m2.load_state_dict(sd)
            m2.train()
            with torch.no_grad():
                m2.step_count.fill_(100)
                la = torch.full((H,), 95, dtype=torch.long)
                dead_idx = np.unique((synth.counter_u64(200, 23, 50) % np.uint64(H)).astype(np.int64))[:11]
                la[torch.from_numpy(dead_idx)] = 3
                m2.feature_last_activated.copy_(la)
            ret = m2.resample_dead_features(torch.from_numpy(x), torch.from_numpy(tgt))
            s2 = sd_numpy(m2)
            out.update({"rs.ret": np.int64(ret), "rs.dead_idx": dead_idx, "rs.W_e": s2["encoder.weight"],
                        "rs.b_e": s2["encoder.bias"], "rs.W_d": s2["decoder.weight"],
                        "rs.last_activated": s2["feature_last_activated"], "rs.step_count": s2["step_count"]})
            # skip transcoder with non-zero decoder / skip weights
            sk = SkipTranscoder(Din, Dout, H, k=K)
            ssd = sk.state_dict()
            ssd.update({k_: v for k_, v in sd.items() if k_ in ssd})
            ws = (synth.normal((Dout, Din), 23, 35) * np.float32(0.05)).astype(np.float32)
            bs = synth.uniform((Dout,), 23, 36, -0.05, 0.05)
            ssd["skip.weight"], ssd["skip.bias"] = torch.from_numpy(ws), torch.from_numpy(bs)
            sk.load_state_dict(ssd)
            sk.train()
            xt = torch.from_numpy(x).requires_grad_(True)
            o = sk(xt, torch.from_numpy(tgt))
            o.loss.backward()
            out.update({"skip.W_s": ws, "skip.b_s": bs, "skip.pred": o.predicted.detach().numpy(),
                        "skip.loss": np.float32(o.loss.item()), "skip.dW_e": sk.encoder.weight.grad.numpy(),
                        "skip.dW_d": sk.decoder.weight.grad.numpy(), "skip.db_d": sk.decoder.bias.grad.numpy(),
                        "skip.dW_s": sk.skip.weight.grad.numpy(), "skip.db_s": sk.skip.bias.grad.numpy(),
                        "skip.dx": xt.grad.numpy(),
                        "skip.contribution": np.float64(sk.get_skip_contribution(torch.from_numpy(x), torch.from_numpy(tgt)))})
    np.savez_compressed(HERE / "g12_transcoders.npz", **out)


def g13_feature_topk():
    """analysis/feature_viz.py:59-250: TopKTracker fed three batches (dense [B,H], sequence [B,T,H], dense again)
    of sparse positive activations without equal values; stored: the inputs and every feature's kept
    (value, sample, position) list, strongest first, plus the counters."""
    from whisper_sae.analysis.feature_viz import TopKTracker  # reference
    H, KEEP = 96, 5
    shapes = [(40, 1), (6, 7), (33, 1)]
    tracker = TopKTracker(num_features=H, k=KEEP)
    out = {"H": np.array(H), "keep": np.array(KEEP)}
    sample0 = 0
    for u, (b, t) in enumerate(shapes):
        n = b * t * H
        raw = synth.counter_u64(n, 13, 10 + u)
        # distinct positive values (the counter index breaks every tie), about one entry in five active
        val = ((raw >> np.uint64(40)).astype(np.float64) + 1.0) / float(1 << 24) + np.arange(n) * 2.0 ** -30
        act = np.where((raw % np.uint64(5)) == 0, val, 0.0).astype(np.float32).reshape(b, t, H)
        assert len(np.unique(act[act > 0])) == int((act > 0).sum())
        samples = list(range(sample0, sample0 + b))
        sample0 += b
        tracker.update(torch.from_numpy(act[:, 0] if t == 1 else act), samples,
                       transcriptions=[f"utt{s}" for s in samples])
        out[f"act{u}"] = act
        out[f"samples{u}"] = np.array(samples)
    vals = np.zeros((H, KEEP), np.float32)
    samp = np.full((H, KEEP), -1, np.int64)
    pos = np.full((H, KEEP), -1, np.int64)
    cnt = np.zeros(H, np.int32)
    for f in range(H):
        ex = tracker.get_top_examples(f)
        cnt[f] = len(ex)
        for j, e in enumerate(ex):
            vals[f, j], samp[f, j], pos[f, j] = e.activation_value, e.sample_idx, e.position_idx
            assert e.transcription == f"utt{e.sample_idx}" and e.timestamp_ms == e.position_idx * 10.0
    out.update(vals=vals, samples=samp, positions=pos, counts=cnt, total_activations=np.array(tracker.total_activations),
               samples_processed=np.array(tracker.samples_processed))
    st = tracker.get_feature_stats()
    out["stats_mean"] = np.array([st[f]["mean_activation"] for f in range(H)], np.float64)
    np.savez_compressed(HERE / "g13_feature_topk.npz", **out)


def tiny_whisper(seed: int = 0):
    """A random-init Whisper small enough for a fixture (no pretrained weights exist offline); same code in
    tests/test_hooks.py builds the same model from the same seed."""
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(vocab_size=200, num_mel_bins=80, encoder_layers=2, decoder_layers=2, encoder_attention_heads=2,
                        decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128, d_model=64,
                        max_source_positions=50, max_target_positions=16, decoder_start_token_id=1, pad_token_id=0,
                        bos_token_id=1, eos_token_id=2)
    torch.manual_seed(seed)
    return WhisperForConditionalGeneration(cfg).eval()


def g14_hooks():
    """sae/hooks.py:147-230: extract_features_batch (with and without the final LayerNorm) on the seeded tiny model."""
    import transformers
    from whisper_sae.sae.hooks import extract_features_batch, flatten_activations  # reference
    model = tiny_whisper(0)
    mel = synth.normal((2, 80, 100), 14, 1).astype(np.float32)
    out = {"mel": mel, "transformers_version": np.array(transformers.__version__), "torch_version": np.array(torch.__version__)}
    for tag, ln in (("ln", True), ("raw", False)):
        r = extract_features_batch(model, torch.from_numpy(mel), [0, 1], [0, 1], ln, "cpu")
        for comp in ("encoder", "decoder"):
            for layer, t in r[comp].items():
                out[f"{tag}.{comp}.{layer}"] = t.numpy()
        out[f"{tag}.flat"] = flatten_activations(r["encoder"][1], "encoder").numpy()
    np.savez_compressed(HERE / "g14_hooks.npz", **out)


def g15_crosscoder():
    """TopKCrossLayerCrosscoder (sae/crosscoder.py:286-379): the seeded initialisation, one training-mode forward with
    every gradient, the dead-feature clock, and the decoder-norm helpers."""
    from whisper_sae.sae.crosscoder import TopKCrossLayerCrosscoder  # reference
    d, L, S, K, B = 32, 3, 256, 8, 40
    layers = [0, 2, 3]
    torch.manual_seed(42)
    m = TopKCrossLayerCrosscoder(d_model=d, n_layers=L, d_sae=S, k=K, layer_indices=layers, dead_feature_threshold=20)
    out = {"dims": np.array([d, L, S, K, B]), "layers": np.array(layers),
           "init.W_dec": m.W_dec.detach().numpy().copy(), "init.W_enc_digest": tensor_digest(m.W_enc.detach().numpy())}
    with torch.no_grad():  # non-trivial biases so their gradients / use are exercised
        m.b_enc.copy_(torch.from_numpy(synth.uniform((S,), 29, 1, -0.05, 0.05)))
        m.b_dec.copy_(torch.from_numpy(synth.uniform((L, d), 29, 2, -0.05, 0.05)))
        m.W_enc.add_(torch.from_numpy((synth.normal((L, d, S), 29, 3) * np.float32(0.02)).astype(np.float32)))
    acts = {li: synth.activations(B, d, seed=29, stream=10 + i, bf16=False) for i, li in enumerate(layers)}
    m.train()
    o = m({li: torch.from_numpy(a) for li, a in acts.items()})
    o.loss.backward()
    out.update({"W_enc": m.W_enc.detach().numpy().copy(), "b_enc": m.b_enc.detach().numpy().copy(),
                "b_dec": m.b_dec.detach().numpy().copy(),
                "acts": np.stack([acts[li] for li in layers]),
                "recon": np.stack([o.reconstructed[li].detach().numpy() for li in layers]),
                "per_layer_loss": np.array([o.per_layer_loss[li].item() for li in layers], dtype=np.float32),
                "loss": np.float32(o.loss.item()), "l0": np.float32(o.l0.item()),
                "idx": np.sort(np.argsort(-o.hidden.detach().numpy(), axis=1, kind="stable")[:, :K], axis=1).astype(np.int16),
                "nnz": (o.hidden.detach().numpy() > 0).sum(axis=1).astype(np.int16),
                "dW_enc": m.W_enc.grad.numpy(), "db_enc": m.b_enc.grad.numpy(), "dW_dec": m.W_dec.grad.numpy(),
                "db_dec": m.b_dec.grad.numpy(), "step_count": np.int64(m.step_count.item()),
                "last_activated": m.feature_last_activated.numpy().copy(),
                "decoder_norms": m.get_decoder_norms().detach().numpy(),
                "layer_norms": m.get_feature_layer_norms().detach().numpy(),
                "cross_layer": m.get_cross_layer_features(0.5).numpy()})
    # encode from a subset of the layers (the others contribute nothing, crosscoder.py:331-336)
    sub = m.encode({2: torch.from_numpy(acts[2])}).detach().numpy()
    out["subset.nnz_idx"] = np.sort(np.argsort(-sub, axis=1, kind="stable")[:, :K], axis=1).astype(np.int16)
    out["subset.hidden_sum"] = sub.sum(axis=1)
    m.normalize_decoder_weights()
    out["normalized.W_dec"] = m.W_dec.detach().numpy().copy()
    np.savez_compressed(HERE / "g15_crosscoder.npz", **out)


def g16_crosscoder_relu():
    """CrossLayerCrosscoder with activation "relu" (sae/crosscoder.py:38-283): one training-mode forward on non-unit
    decoder norms (so the norm weights and their own gradient matter), every gradient, the clock."""
    from whisper_sae.sae.crosscoder import CrossLayerCrosscoder  # reference
    d, L, S, B, lam = 32, 3, 256, 40, 0.05
    layers = [1, 2, 5]
    torch.manual_seed(7)
    m = CrossLayerCrosscoder(d_model=d, n_layers=L, d_sae=S, layer_indices=layers, sparsity_weight=lam, dead_feature_threshold=20)
    with torch.no_grad():
        m.b_enc.copy_(torch.from_numpy(synth.uniform((S,), 31, 1, -0.05, 0.05)))
        m.b_dec.copy_(torch.from_numpy(synth.uniform((L, d), 31, 2, -0.05, 0.05)))
        m.W_enc.add_(torch.from_numpy((synth.normal((L, d, S), 31, 3) * np.float32(0.02)).astype(np.float32)))
        m.W_dec.mul_(torch.from_numpy(synth.uniform((S, 1, 1), 31, 4, 0.5, 2.0)))   # decoder norms 0.05 .. 0.2
    acts = {li: synth.activations(B, d, seed=31, stream=10 + i, bf16=False) for i, li in enumerate(layers)}
    m.train()
    o = m({li: torch.from_numpy(a) for li, a in acts.items()})
    o.loss.backward()
    out = {"dims": np.array([d, L, S, B]), "layers": np.array(layers), "lam": np.float64(lam),
           "W_enc": m.W_enc.detach().numpy().copy(), "b_enc": m.b_enc.detach().numpy().copy(),
           "W_dec": m.W_dec.detach().numpy().copy(), "b_dec": m.b_dec.detach().numpy().copy(),
           "acts": np.stack([acts[li] for li in layers]),
           "recon": np.stack([o.reconstructed[li].detach().numpy() for li in layers]),
           "hidden_digest": tensor_digest(o.hidden.detach().numpy()), "hidden_rowsum": o.hidden.detach().numpy().sum(axis=1),
           "per_layer_loss": np.array([o.per_layer_loss[li].item() for li in layers], dtype=np.float32),
           "loss": np.float32(o.loss.item()), "reconstruction_loss": np.float32(o.reconstruction_loss.item()),
           "sparsity_loss": np.float32(o.sparsity_loss.item()), "l0": np.float32(o.l0.item()),
           "dW_enc": m.W_enc.grad.numpy(), "db_enc": m.b_enc.grad.numpy(), "dW_dec": m.W_dec.grad.numpy(),
           "db_dec": m.b_dec.grad.numpy(), "step_count": np.int64(m.step_count.item()),
           "last_activated": m.feature_last_activated.numpy().copy()}
    np.savez_compressed(HERE / "g16_crosscoder_relu.npz", **out)


def g17_extraction_driver():
    """data/feature_cache.py:200-306: extract_and_cache_features on the seeded tiny Whisper - three mel batches of two
    clips, max_samples = 5 (whole batches are taken while fewer than 5 samples are in: the count ends at 6), encoder layers
    0 and 1 - into a FeatureCache.  Stored: every cached tensor and every field of its metadata sidecar.  (Decoder layers:
    with the transformers build of this image a decoder layer returns a bare tensor, the reference's hook takes output[0] of it
    (hooks.py:99-101) and its driver then raises ValueError in flatten_activations - recorded as ``decoder_raises``.)"""
    import contextlib
    import io
    import transformers
    from whisper_sae.config import DataConfig, WhisperConfig  # reference
    from whisper_sae.data.feature_cache import FeatureCache, extract_and_cache_features  # reference
    model = tiny_whisper(0)
    mel = synth.normal((8, 80, 100), 17, 1).astype(np.float32)
    batches = [torch.from_numpy(mel[i:i + 2]) for i in range(0, 8, 2)]
    out = {"mel": mel, "transformers_version": np.array(transformers.__version__), "torch_version": np.array(torch.__version__)}
    with tempfile.TemporaryDirectory() as td:
        fc = FeatureCache(Path(td), WhisperConfig(), DataConfig(cache_dir=Path("cache")))
        with contextlib.redirect_stdout(io.StringIO()):
            extract_and_cache_features(model, None, batches, fc, [0, 1], [], device="cpu", max_samples=5)
            try:
                extract_and_cache_features(model, None, batches[:1], fc, [], [1], device="cpu", max_samples=2)
                out["decoder_raises"] = np.array("")
            except Exception as exc:  # noqa: BLE001
                out["decoder_raises"] = np.array(type(exc).__name__)
        files = sorted(p.name for p in Path(td).iterdir())
        out["files"] = np.array(files)
        for comp, layer in (("encoder", 0), ("encoder", 1)):
            feats, meta = fc.load(comp, layer)
            out[f"{comp}.{layer}"] = feats.numpy()
            md = json.loads(meta.to_json()) if hasattr(meta, "to_json") else dict(meta.__dict__)
            md["created_at"] = ""
            out[f"meta.{comp}.{layer}"] = np.array(json.dumps(md, sort_keys=True, default=str))
    np.savez_compressed(HERE / "g17_extraction.npz", **out)


SETS = {"g17": g17_extraction_driver, "g16": g16_crosscoder_relu, "g15": g15_crosscoder, "g14": g14_hooks, "g13": g13_feature_topk, "g12": g12_transcoders, "g1": g1_g2_g3, "g4": g4_trajectory, "g4b": g4b_trajectory_cfg2, "g5": g5_lr, "g6": g6_dead, "g7": g7_resample, "g8": g8_relu,
        "g10": g10_seeded_init, "g11": g11_cache_interchange}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(SETS)):
        SETS[name]()
    for p in sorted(HERE.glob("g*")):
        print(p.name, p.stat().st_size)
