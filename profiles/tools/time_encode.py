"""Event-timed averages of the encoder GEMM and the strip TopK at cfg 2 for the library WSAE_LIB points at."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "whisper-sae_amd")]
from oracle import synth
from whisper_sae import _native as N
from whisper_sae.sae.model import TopKSAE
D, H, K, Bb = 384, 3072, 32, 16384
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = TopKSAE(D, H, k=K, precision="bf16").to(dev)
eng = m.bind()
lib = N.lib()
xb = torch.from_numpy(synth.activations(Bb, D, seed=1, stream=0, bf16=True)).to(dev).to(torch.bfloat16)
h = eng.prepare(N.PREC_BF16, Bb, force=True)
v = torch.empty(Bb, K, device=dev); i = torch.empty(Bb, K, dtype=torch.int32, device=dev)
stn = eng.stats.data_ptr(); s = eng.stream(); pk = eng.pack.data_ptr()
def run(n):
    for _ in range(n):
        N.check(lib.wsae_encode_topk(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, stn, s), "x")
    torch.cuda.synchronize()
run(50)
N.check(lib.wsae_profile_enable(h, -1, 4096), "prof")
run(100)
p = N.profile_read(h)
print(sys.argv[1] if len(sys.argv) > 1 else "", {k.split("<")[0]: round(ms / n * 1e3, 1) for k, (n, ms) in p.items()})
