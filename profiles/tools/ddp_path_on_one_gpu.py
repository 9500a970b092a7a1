"""What the data-parallel STRUCTURE of the step costs without any communication: one process, a one-rank NCCL (RCCL) group,
`world()` patched to claim two ranks so that the trainer takes its DDP branch (two-half backward onto the wire, three
all-reduce calls - no-ops on one rank -, wire unpack).  Prints the plain step and the DDP-structured step, same box.
    python profiles/tools/ddp_path_on_one_gpu.py [fp32|bf16] [halves]"""
import os, sys, time
from pathlib import Path
import torch, torch.distributed as dist
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "whisper-sae_amd")]
from whisper_sae.config import TrainingConfig
from whisper_sae.data import ActivationRing, RingLoader
from whisper_sae.sae.model import TopKSAE
from whisper_sae.sae.training import SAETrainer
import whisper_sae.distributed as D
import whisper_sae.sae.training as T

wire = sys.argv[1] if len(sys.argv) > 1 else "fp32"
halves = len(sys.argv) > 2 and sys.argv[2] == "halves"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B, steps = 16384, 200

def run(ddp: bool) -> float:
    torch.manual_seed(42)
    model = TopKSAE(384, 3072, k=32)
    cfg = TrainingConfig(batch_size=B, learning_rate=1e-4, weight_decay=0.0, warmup_steps=1000, gradient_clip=1.0, use_amp=True,
                         num_workers=0, seed=42, grad_exchange_dtype=wire, ddp_overlap_halves=halves)
    tr = SAETrainer(model, cfg, device=dev, run_dir=ROOT / "gpurun_out" / "ddp_probe")
    ring = ActivationRing(1 << 22, 384, device=dev, dtype=torch.bfloat16)
    ring.fill_synthetic(1 << 22, seed=42)
    loader = RingLoader(ring, B, shuffle=True, seed=42)

    def batches():
        while True:
            for b in loader:
                if len(b) == B:
                    yield b

    it = batches()
    tr.setup_scheduler(20000)
    for _ in range(100):
        tr.train_step(next(it))
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step(next(it))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    # per-kernel HIP-event times of 50 more steps (the events add a few us per launch)
    from whisper_sae import _native as N
    eng = model._engine
    handle = eng.ctx(N.PREC_BF16, B)
    N.check(N.lib().wsae_profile_enable(handle, -1, 50), "wsae_profile_enable")
    for _ in range(50):
        tr.train_step(next(it))
    torch.cuda.synchronize()
    pr = N.profile_read(handle)
    N.check(N.lib().wsae_profile_disable(handle), "wsae_profile_disable")
    print("   kernels (us per step, calls per step):", {k: (round(v[1] / 50 * 1e3, 1), v[0] / 50) for k, v in pr.items() if v[0]})
    del tr, ring
    return best * 1e6

plain = run(False)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
fake = lambda: (dist, 2)
D.world = fake
T.world = fake
if os.environ.get("NO_COLLECTIVES"):  # experiment: the structure alone (wire + unpack), no torch.distributed call at all
    class _Ex0(D.WireExchange):
        def start(self, view):
            pass
        def run(self, view):
            pass
    T.WireExchange = _Ex0
elif os.environ.get("NO_METRIC_COLLECTIVE"):  # experiment: how much of the overhead is the 8-byte metric all-reduce
    class _Ex(D.WireExchange):
        def start(self, view):
            if view.numel() > 2:
                super().start(view)
    T.WireExchange = _Ex
ddp = run(True)
print(f"plain step {plain:.1f} us; DDP-structured step on one rank ({wire} wire, {'two halves' if halves else 'one launch'}, no communication) {ddp:.1f} us: +{ddp - plain:.1f} us")
dist.destroy_process_group()
