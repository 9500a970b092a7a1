"""Torch-CPU restatement of one reference train step, used for the ``cpu_baseline`` leg of bench.py
and as a second opinion next to the numpy oracle.

TEST INFRASTRUCTURE (see oracle/sae_oracle.py header).  This is the op sequence the reference issues
per step on its CPU path -- ``SAETrainer.train_step`` (training.py:161-217) -> ``TopKSAE.forward``
(model.py:131-166): fp32, AMP off (training.py:73-75), autograd backward, ``clip_grad_norm_``,
``torch.optim.AdamW``, ``F.normalize(decoder.weight, dim=0)``, dead-feature bookkeeping and the five
``.item()`` host reads -- written functionally on plain tensors (kind = "port": the reference's
Python files do not travel to the GPU box).  tests/test_oracle_golden.py checks it against the same
golden vectors as the numpy oracle.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


class TorchCPUStep:
    def __init__(self, weights: dict, k: int, lr: float = 1e-4, weight_decay: float = 0.0,
                 max_norm: float = 1.0, dead_feature_threshold: int = 10_000):
        t = lambda a: torch.tensor(a, dtype=torch.float32).clone().requires_grad_(True)  # noqa: E731
        self.W_e, self.b_e = t(weights["encoder.weight"]), t(weights["encoder.bias"])
        self.W_d, self.b_d = t(weights["decoder.weight"]), t(weights["decoder.bias"])
        self.b_pre = t(weights["b_pre"])
        self.k, self.max_norm, self.thr = k, max_norm, dead_feature_threshold
        self.params = [self.b_pre, self.W_e, self.b_e, self.W_d, self.b_d]  # module.parameters() order
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay)
        self.last_activated = torch.zeros(self.W_e.shape[0], dtype=torch.long)
        self.step_count = torch.tensor(0, dtype=torch.long)

    def forward(self, x: torch.Tensor):
        pre = F.linear(x - self.b_pre, self.W_e, self.b_e)  # model.py:108-111
        vals, idx = torch.topk(pre, self.k, dim=-1)  # model.py:114
        hidden = torch.zeros_like(pre).scatter_(-1, idx, torch.relu(vals))  # model.py:115-116
        recon = F.linear(hidden, self.W_d, self.b_d) + self.b_pre  # model.py:129
        loss = F.mse_loss(recon, x)  # model.py:145
        l0 = (hidden > 0).float().sum(dim=-1).mean()  # model.py:148
        self.step_count += 1  # model.py:174-181
        self.last_activated[(hidden > 0).any(dim=0)] = self.step_count
        return recon, hidden, loss, l0

    def step(self, x: torch.Tensor, lr: float | None = None) -> dict:
        if lr is not None:
            for g in self.opt.param_groups:
                g["lr"] = lr
        _, _, loss, l0 = self.forward(x)
        self.opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)  # training.py:188-191
        self.opt.step()  # training.py:193
        with torch.no_grad():  # training.py:197-198 -> model.py:91-96
            self.W_d.data = F.normalize(self.W_d.data, dim=0)
        dead = ((self.step_count - self.last_activated) > self.thr).float().mean()
        # the reference reads five scalars back per step (training.py:207-213)
        return {"loss": loss.item(), "reconstruction_loss": loss.item(), "sparsity_loss": 0.0, "l0": l0.item(),
                "dead_feature_ratio": dead.item(), "grad_norm": float(gn)}
