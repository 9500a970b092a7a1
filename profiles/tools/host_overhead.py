"""Host time of one SAETrainer.train_step call against the device time of its kernels, at small batches (where the step is
launch- or host-bound).  python profiles/tools/host_overhead.py [B ...]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "whisper-sae_amd")]
import torch  # noqa: E402

from whisper_sae.config import TrainingConfig  # noqa: E402
from whisper_sae.data import ActivationRing, RingLoader  # noqa: E402
from whisper_sae.sae.model import TopKSAE  # noqa: E402
from whisper_sae.sae.training import SAETrainer  # noqa: E402

for B in [int(a) for a in sys.argv[1:]] or [128, 4096]:
    torch.manual_seed(42)
    dev = torch.device("cuda", 0)
    model = TopKSAE(384, 3072, k=32)
    cfg = TrainingConfig(batch_size=B, learning_rate=1e-4, warmup_steps=1000, use_amp=True, num_workers=0)
    tr = SAETrainer(model, cfg, device=dev, run_dir=ROOT / "gpurun_out" / "host_overhead")
    ring = ActivationRing(1 << 18, 384, device=dev, dtype=torch.bfloat16)
    ring.fill_synthetic(1 << 18, seed=1)
    tr.setup_scheduler(100000)
    it = iter(RingLoader(ring, B, shuffle=True, seed=1))
    batches = [next(it) for _ in range(600)]
    for b in batches[:100]:
        tr.train_step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[100:600]:
        tr.train_step(b)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"B={B}: host {t_host / 500 * 1e6:.1f} us/step enqueue, {t_all / 500 * 1e6:.1f} us/step with the device drained")
    pr = cProfile.Profile()
    pr.enable()
    for b in batches[:300]:
        tr.train_step(b)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(18)
