"""Data-parallel path on CPU: world_size 2, gloo backend (the GPU path runs the same code on RCCL).

Each rank computes the oracle's gradients on its own half batch, packs them in the flat layout the
kernels use, and calls ``whisper_sae.distributed.sync_gradients`` (the function the trainer calls
between ``wsae_weight_grads`` and ``wsae_adamw_step``) on the buffer ``[gradients | fired]``.  The
result, scaled by the returned factor, must equal the oracle's gradients on the concatenated batch, and
the dead-feature clocks merged from the summed ``fired`` indicators (what ``wsae_adamw_step`` does on
the device, restated by ``merge_clock``) must equal the single-process clocks -- i.e. DDP over N ranks
is the single-device step on N*B rows, with ONE collective.
"""

from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sae_oracle as O
from oracle import synth
from whisper_sae import _native as N
from whisper_sae.distributed import merge_clock, rank_and_world, sync_gradients

D, H, K, B = 64, 256, 8, 32
PACK_ORDER = ("W_e", "W_d", "b_e", "b_d", "b_pre")


def pack(grads: dict) -> torch.Tensor:
    total, off = N.pack_layout(D, H)
    flat = np.zeros(total, dtype=np.float32)
    flat[off[0]:off[1]] = grads["W_e"].reshape(-1)
    flat[off[1]:off[2]] = grads["W_d"].T.reshape(-1)  # W_dT layout
    flat[off[2]:off[3]] = grads["b_e"]
    flat[off[3]:off[4]] = grads["b_d"]
    flat[off[4]:] = grads["b_pre"]
    return torch.from_numpy(flat)


def _worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert rank_and_world() == (rank, world)
        w = synth.sae_weights(D, H, seed=3, bf16=False, b_pre_scale=0.05)
        st = O.SAEState.from_state_dict(w, k=K, dead_feature_threshold=5)
        clocks_before = st.last_activated.copy()
        x = synth.activations(world * B, D, seed=3, stream=8, bf16=False)
        mine = x[rank * B:(rank + 1) * B]
        fwd = O.forward(st, mine, "fp32", training=True)
        grads = pack(O.backward(st, mine, fwd, "fp32"))
        step = int(st.step_count)
        local = torch.from_numpy(st.last_activated.copy())
        fired = (local == step).to(torch.float32)  # what the decode kernel stores: 1.0 where this rank stamped the clock
        last = torch.from_numpy(clocks_before.copy())  # every rank starts the step with the agreed clocks
        flat = torch.cat([grads, fired])
        wire = flat.clone()
        scale = sync_gradients(flat)
        assert scale == 1.0 / world
        # the optional bf16 exchange (TrainingConfig.grad_exchange_dtype = "bf16"): the same sum to bf16 accuracy,
        # the fired indicators exactly
        assert sync_gradients(wire, torch.bfloat16) == scale
        np.save(os.path.join(out_dir, f"w{rank}.npy"), wire.numpy())
        last = merge_clock(last, flat[grads.numel():], step)
        flat = flat[:grads.numel()]
        np.save(os.path.join(out_dir, f"g{rank}.npy"), (flat * scale).numpy())
        np.save(os.path.join(out_dir, f"l{rank}.npy"), last.numpy())
    finally:
        dist.destroy_process_group()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
def test_two_rank_gradient_average_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w = synth.sae_weights(D, H, seed=3, bf16=False, b_pre_scale=0.05)
    st = O.SAEState.from_state_dict(w, k=K, dead_feature_threshold=5)
    x = synth.activations(world * B, D, seed=3, stream=8, bf16=False)
    fwd = O.forward(st, x, "fp32", training=True)
    want = pack(O.backward(st, x, fwd, "fp32")).numpy()
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)  # every rank holds the same reduced gradients
    assert np.abs(g0 - want).max() <= 2e-6 * np.abs(want).max()
    l0, l1 = np.load(tmp_path / "l0.npy"), np.load(tmp_path / "l1.npy")
    assert np.array_equal(l0, l1) and np.array_equal(l0, st.last_activated)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    n = want.size
    assert np.array_equal(w0, w1)
    assert np.abs(w0[:n] / world - want).max() <= 2.0 ** -7 * np.abs(want).max()
    assert np.array_equal(w0[n:] > 0, l0 == int(st.step_count))  # fired sums survive the bf16 wire exactly


def test_single_process_is_a_no_op():
    flat = torch.arange(8, dtype=torch.float32)
    assert sync_gradients(flat) == 1.0 and torch.equal(flat, torch.arange(8, dtype=torch.float32))
    last = torch.tensor([3, 0, 7], dtype=torch.int64)
    assert merge_clock(last, torch.tensor([0.0, 2.0, 0.0]), 9).tolist() == [3, 9, 7]
    assert rank_and_world() == (0, 1)


# ---- the wire protocol of the product step (include/wsae.h), four ranks ---------------------------------------------------
def _oracle_wire(st, x_mine, mode, dtype):
    """One rank's contribution to the wire: the oracle's gradients in pack order + fired, laid out as the reduction kernel
    writes them."""
    from whisper_sae.distributed import pack_to_wire
    before = st.last_activated.copy()
    fwd = O.forward(st, x_mine, mode, training=True)
    grads = pack(O.backward(st, x_mine, fwd, mode))
    fired = torch.from_numpy((st.last_activated == int(st.step_count)).astype(np.float32))
    st.last_activated = before  # the clock is merged from the summed indicators below
    return pack_to_wire(torch.cat([grads, fired]), D, H, dtype, metrics=(float(fwd["loss"]), float(fwd["l0"]))), fwd


def _unpack_grads(flat: torch.Tensor) -> dict:
    total, off = N.pack_layout(D, H)
    f = flat.numpy()
    return {"W_e": f[off[0]:off[1]].reshape(H, D), "W_d": f[off[1]:off[2]].reshape(H, D).T.copy(), "b_e": f[off[2]:off[3]],
            "b_d": f[off[3]:off[4]], "b_pre": f[off[4]:total]}


def _worker_wire(rank: int, world: int, port: int, out_dir: str, steps: int):
    from whisper_sae.distributed import WIRE_METRIC_SLOTS, WireExchange, decode_wire_metrics, wire_offsets, wire_to_pack
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total, _ = N.pack_layout(D, H)
        wo = wire_offsets(D, H)
        assert wo["total"] == total + H + WIRE_METRIC_SLOTS and wo["fired"] == total and wo["metrics"] == total + H
        xs = synth.activations(steps * world * B, D, seed=3, stream=8, bf16=False).reshape(steps, world * B, D)
        for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            w = synth.sae_weights(D, H, seed=3, bf16=False, b_pre_scale=0.05)
            st = O.SAEState.from_state_dict(w, k=K, dead_feature_threshold=5)
            losses = []
            for s in range(steps):
                wire, fwd = _oracle_wire(st, xs[s, rank * B:(rank + 1) * B], "fp32", dtype)
                ex = WireExchange()
                if s % 2 == 0:                     # the two-half form ...
                    ex.start(wire[:wo["split"]])   # the decoder half goes first ...
                    ex.start(wire[wo["split"]:])   # ... the rest follows (the trainer launches the encoder half in between)
                else:                              # ... and the default: one collective, in stream order
                    ex.run(wire)
                scale = ex.finish()
                # the step's (loss, l0) came over the wire as digits: their sums are exact whatever the wire dtype
                met = decode_wire_metrics(wire[wo["metrics"]:].float(), 1)
                ref = torch.tensor([float(fwd["loss"]), float(fwd["l0"])], dtype=torch.float64)
                dist.all_reduce(ref)
                assert abs(met[0] - float(ref[0])) <= 1e-6 * world and abs(met[1] - float(ref[1])) <= 2e-5 * world, (met, ref)
                flat = wire_to_pack(wire, D, H)
                step_now = int(st.step_count)
                st.last_activated = merge_clock(torch.from_numpy(st.last_activated), flat[total:], step_now).numpy()
                red = _unpack_grads(flat[:total] * scale)
                st_step, st_last = int(st.step_count), st.last_activated.copy()
                O.train_step(st, xs[s, rank * B:(rank + 1) * B], 1e-3, "fp32", max_norm=1.0, reduced_grads=red)
                # the step's own forward advanced the clock a second time: keep the exchange's view of it
                st.step_count, st.last_activated = np.int64(st_step), st_last
                losses.append(met[0] * scale)
            np.savez(os.path.join(out_dir, f"{tag}_{rank}.npz"), W_e=st.W_e, W_d=st.W_d, b_e=st.b_e, b_d=st.b_d, b_pre=st.b_pre,
                     last=st.last_activated, losses=np.array(losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_four_ranks_on_the_wire_in_two_halves(tmp_path):
    """World 4 over gloo: the wire layout, the two asynchronous half collectives + the metric pair, the clock merge - on the
    fp32 wire the trajectory is the single-process one on the concatenated batches; the bf16 wire (opt-in) stays inside a
    band around it (a multi-step trajectory test above world 2 for the rounded exchange)."""
    world, steps = 4, 6
    mp.spawn(_worker_wire, args=(world, _free_port(), str(tmp_path), steps), nprocs=world, join=True)
    f = [np.load(tmp_path / f"f32_{r}.npz") for r in range(world)]
    b = [np.load(tmp_path / f"bf16_{r}.npz") for r in range(world)]
    for r in range(1, world):
        for key in f[0].files:
            assert np.array_equal(f[0][key], f[r][key]), ("fp32 wire", key, r)   # every rank holds the same state
            assert np.array_equal(b[0][key], b[r][key]), ("bf16 wire", key, r)
    # single process on the concatenated batches
    w = synth.sae_weights(D, H, seed=3, bf16=False, b_pre_scale=0.05)
    st = O.SAEState.from_state_dict(w, k=K, dead_feature_threshold=5)
    xs = synth.activations(steps * world * B, D, seed=3, stream=8, bf16=False).reshape(steps, world * B, D)
    losses = [O.train_step(st, xs[s], 1e-3, "fp32", max_norm=1.0)["loss"] for s in range(steps)]
    assert np.allclose(f[0]["losses"], losses, rtol=1e-6)                      # mean of the ranks' batch means
    assert np.array_equal(f[0]["last"], st.last_activated)
    for key, ref in (("W_e", st.W_e), ("W_d", st.W_d), ("b_e", st.b_e), ("b_pre", st.b_pre)):
        d = np.abs(f[0][key].astype(np.float64) - ref.astype(np.float64))
        assert np.mean(d < 0.05 * 1e-3) > 0.995, key     # entries whose gradient is at rounding level may step the other way
        assert d.max() < 2.5e-3 * steps, key
        # the rounded wire: 2^-8 relative per addend on the gradients; Adam turns that into sign flips of near-zero entries only
        db = np.abs(b[0][key].astype(np.float64) - f[0][key].astype(np.float64))
        assert np.mean(db < 0.05 * 1e-3) > 0.95, key  # measured: 0.965 (W_d) .. 0.99
        assert db.max() < 5e-3 * steps, key           # measured: 0.019 on W_d (unit-norm columns: entries O(0.1), renormalised every step)
    assert np.array_equal(b[0]["last"], f[0]["last"])                          # the indicators survive bf16 exactly
    assert np.allclose(b[0]["losses"], f[0]["losses"], rtol=2e-3)


def test_wire_layout_round_trip():
    from whisper_sae.distributed import pack_to_wire, wire_offsets, wire_to_pack
    total, off = N.pack_layout(D, H)
    flat = torch.arange(total + H, dtype=torch.float32)
    wire = pack_to_wire(flat, D, H)
    wo = wire_offsets(D, H)
    assert wire[wo["W_dT"]] == flat[off[1]] and wire[wo["W_e"]] == flat[off[0]] and wire[wo["b_e"]] == flat[off[2]]
    assert wire[wo["fired"]] == flat[total] and wire.numel() == wo["total"]
    assert torch.equal(wire_to_pack(wire, D, H), flat)


def test_wire_metric_digits_sum_exactly_in_bf16():
    """(loss, l0) as base-16 digits: the digit sums of up to 16 ranks stay below 256, which bf16 holds exactly - summed in any
    order, in bf16, they decode to the fp64 mean of the ranks' values to the fixed-point resolution."""
    from whisper_sae.distributed import WIRE_METRIC_SLOTS, decode_wire_metrics, encode_wire_metrics
    rng = np.random.default_rng(0)
    for world in (1, 2, 8, 16):
        losses = rng.uniform(1e-4, 3.0, world) * rng.choice([1.0, 100.0], world)
        l0s = rng.integers(0, 64 * 16384, world) / 16384.0
        acc = torch.zeros(WIRE_METRIC_SLOTS, dtype=torch.bfloat16)
        for r in rng.permutation(world):
            acc = (acc + torch.tensor(encode_wire_metrics(float(losses[r]), float(l0s[r])), dtype=torch.bfloat16))  # bf16 adds
        loss, l0 = decode_wire_metrics(acc.float(), world)
        assert abs(loss - losses.mean()) <= 2.0 ** -24 and abs(l0 - l0s.mean()) <= 2.0 ** -16, (world, loss, losses.mean())
    bad = torch.tensor(encode_wire_metrics(float("nan"), 1.0)) + torch.tensor(encode_wire_metrics(0.5, 1.0))
    assert all(np.isnan(v) for v in decode_wire_metrics(bad, 2))
