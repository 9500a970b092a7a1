"""Per-step GPU time over the first steps of a fresh process (what the driver's `--steps 20 --warmup 5` run sees)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "whisper-sae_amd")]
import torch

from whisper_sae.config import TrainingConfig
from whisper_sae.data import ActivationRing, RingLoader
from whisper_sae.sae.model import TopKSAE
from whisper_sae.sae.training import SAETrainer

dev = torch.device("cuda:0")
torch.manual_seed(42)
B = 16384
model = TopKSAE(384, 3072, k=32)
tr = SAETrainer(model, TrainingConfig(batch_size=B, learning_rate=1e-4, warmup_steps=1000, use_amp=True, num_workers=0), device=dev,
                run_dir=ROOT / "gpurun_out" / "ramp")
ring = ActivationRing(1 << 22, 384, device=dev, dtype=torch.bfloat16)
ring.fill_synthetic(1 << 22, seed=42)
loader = RingLoader(ring, B, shuffle=True, seed=42)
tr.setup_scheduler(20000)
it = iter(loader)
n = 140
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    tr.train_step(next(it))
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("steps 0-9   :", " ".join(f"{v:.3f}" for v in ms[:10]))
for lo in range(10, n, 10):
    print(f"steps {lo}-{lo+9}: mean {sum(ms[lo:lo+10])/10:.4f} max {max(ms[lo:lo+10]):.3f}")
