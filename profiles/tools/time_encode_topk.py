"""Event-timed encoder GEMM + TopK alone (wsae_encode_topk), for quick experiments on the encoder epilogue."""
import sys, ctypes as C
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
eng = m.bind(); lib = N.lib()
xs = [torch.from_numpy(synth.activations(Bb, D, seed=s, stream=0, bf16=True)).to(dev).to(torch.bfloat16) for s in range(4)]
h = eng.prepare(N.PREC_BF16, Bb, force=True)
if len(sys.argv) > 1:
    N.check(lib.wsae_ctx_set_strip_predict(h, 1, float(sys.argv[1])), "sp")
v = torch.empty(Bb, K, device=dev); i = torch.empty(Bb, K, dtype=torch.int32, device=dev)
stn = eng.stats.data_ptr(); s = eng.stream(); pk = eng.pack.data_ptr()
def run(n):
    for j in range(n):
        if len(sys.argv) > 1:
            N.check(lib.wsae_ctx_set_strip_predict(h, 1, float(sys.argv[1])), "sp")
        N.check(lib.wsae_encode_topk(h, pk, xs[j % 4].data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, stn, s), "x")
run(20); torch.cuda.synchronize()
N.check(lib.wsae_profile_enable(h, -1, 200), "pe")
run(200); torch.cuda.synchronize()
pr = N.profile_read(h)
print({k: round(t[1] / max(t[0], 1) * 1e3, 2) for k, t in pr.items() if t[0]})
