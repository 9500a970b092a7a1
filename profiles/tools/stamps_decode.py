"""Per-phase cycle shares of the MFMA decode kernel's row loop (diagnostic library built with -DWSAE_DM_STAMPS:
profiles/tools/build/libwsae_stamps.so; run with WSAE_LIB pointing at it)."""
import ctypes as C, sys
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
dbg = C.CDLL(str(N.library_path())).wsae_debug_stamps
dbg.argtypes = [C.POINTER(C.c_double), C.c_int]
xb = torch.from_numpy(synth.activations(Bb, D, seed=1, stream=0, bf16=True)).to(dev).to(torch.bfloat16)
h = eng.prepare(N.PREC_BF16, Bb, force=True)
v = torch.empty(Bb, K, device=dev); i = torch.empty(Bb, K, dtype=torch.int32, device=dev); dp = torch.empty(Bb, K, device=dev)
stn = eng.stats.data_ptr(); s = eng.stream(); pk = eng.pack.data_ptr()
names = ["first code", "row head (stamps, x loads, A frag, sources)", "DMA issue x2", "next row's code / TopK", "vm_wait",
         "pass 1 (+residual)", "pass 2 + DMA issue", "row outputs"]
def run(fused, reps=5):
    out = (C.c_double * 8)()
    dbg(out, 1)
    for _ in range(reps):
        if fused:
            N.check(lib.wsae_encode_decode(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, 0, 1, dp.data_ptr(), 0, stn, s), "x")
        else:
            N.check(lib.wsae_encode_topk(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, stn, s), "x")
            N.check(lib.wsae_decode_loss(h, pk, xb.data_ptr(), N.DT_BF16, 0, v.data_ptr(), i.data_ptr(), Bb, 0, 1, dp.data_ptr(), 0, 0, stn, s), "x")
    torch.cuda.synchronize()
    dbg(out, 1)
    rows = Bb * reps
    tot = sum(out)
    print("fused" if fused else "separate", f": {tot / rows:.0f} cycles per row")
    for n_, c in zip(names, out):
        print(f"   {n_:48s} {c / rows:9.1f}  {100 * c / tot:5.1f} %")
run(False)
