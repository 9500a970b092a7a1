"""Per-phase cycle shares of the persistent encoder GEMM (diagnostic library built with -DWSAE_ENC_STAMPS; run with
WSAE_LIB pointing at it):  make -C whisper-sae_amd/csrc OUT=$PWD/build_ab/libwsae_stamps.so BUILD=$PWD/build_ab/obj_st EXTRA=-DWSAE_ENC_STAMPS"""
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
dbg = C.CDLL(str(N.library_path())).wsae_debug_enc_stamps
dbg.argtypes = [C.POINTER(C.c_double), C.c_int]
xb = torch.from_numpy(synth.activations(Bb, D, seed=1, stream=0, bf16=True)).to(dev).to(torch.bfloat16)
h = eng.prepare(N.PREC_BF16, Bb, force=True)
v = torch.empty(Bb, K, device=dev); i = torch.empty(Bb, K, dtype=torch.int32, device=dev)
stn = eng.stats.data_ptr(); s = eng.stream(); pk = eng.pack.data_ptr()
names = ["tile head (acc zero, next rows)", "dma_wait (slab landed)", "barrier (loop top)", "DMA issue", "MFMA slab (reads + 32 MFMA)",
         "barrier (before epilogue)", "epilogue: accumulators -> LDS patch", "epilogue: patch reads, bias, stores, strip max"]
out = (C.c_double * 8)()
reps = 5
for _ in range(3):
    N.check(lib.wsae_encode_topk(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, stn, s), "x")
torch.cuda.synchronize()
dbg(out, 1)
for _ in range(reps):
    N.check(lib.wsae_encode_topk(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, stn, s), "x")
torch.cuda.synchronize()
dbg(out, 1)
waves = 256 * 8 * reps
tot = sum(out)
print(f"{tot / waves:.0f} memtime ticks per wave and launch (100 MHz ticks: x10 ns)")
for n_, c in zip(names, out):
    print(f"   {n_:48s} {c / waves:9.1f}  {100 * c / tot:5.1f} %")
