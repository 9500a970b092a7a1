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
