"""Debug harness for the MFMA decode kernel: element-wise recon / dpre against the oracle on a small batch, with
the error broken down by column position, then timings of the separate and fused launches at the bench batch."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "whisper-sae_amd")]
from oracle import sae_oracle as O, synth
from whisper_sae import _native as N
from whisper_sae.sae.model import TopKSAE

D, H, K = [int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (384, 3072, 32))]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda:0")
w = synth.sae_weights(D, H, seed=42, bf16=False, b_pre_scale=0.1)
m = TopKSAE(D, H, k=K, precision="bf16")
sd = m.state_dict()
for k_ in ("encoder.weight", "encoder.bias", "decoder.weight", "decoder.bias", "b_pre"):
    sd[k_] = torch.from_numpy(w[k_])
m.load_state_dict(sd); m.to(dev).train()
st = O.SAEState.from_state_dict(w, k=K)
x = synth.activations(B, D, seed=42, stream=1, bf16=True)
xt = torch.from_numpy(x).to(dev).requires_grad_(True)
out = m(xt)
out.loss.backward()
vals, idx = m._last_code
fwd = O.forward(st.copy(), x, "amp", select=idx.cpu().numpy())
ora = O.backward(st, x, fwd, "amp")
rec = out.reconstructed.detach().cpu().numpy()
err = np.abs(rec - fwd["reconstructed"])
print("loss", float(out.loss), float(fwd["loss"]), "recon max err", err.max(), "rel", err.max() / np.abs(fwd["reconstructed"]).max())
print("err by column%16 :", np.round(err.max(axis=0).reshape(-1, 16).max(axis=0), 6))
print("err by column//128:", np.round(err.max(axis=0).reshape(-1, 128).max(axis=1), 6))
print("err by row (first 16):", np.round(err.max(axis=1)[:16], 6))
# which decomposition explains the error? recon without lo parts etc
hid = fwd["hidden"].astype(np.float64)
wd = synth.bf16_round(st.W_d).astype(np.float64)
hi = synth.bf16_round(hid.astype(np.float32)).astype(np.float64)
for name, hh in (("hi only", hi), ("exact", hid)):
    r = hh @ wd.T + st.b_d + st.b_pre
    print(name, "max |kernel - variant|", np.abs(rec - r).max())
eng = m._engine
g = {"W_e": m.encoder.weight.grad, "b_e": m.encoder.bias.grad, "W_d": m.decoder.weight.grad, "b_d": m.decoder.bias.grad,
     "b_pre": m.b_pre.grad}
for n, t in g.items():
    a = t.detach().cpu().numpy().astype(np.float64); b = ora[n].astype(np.float64)
    print("grad", n, "rel err", np.abs(a - b).max() / np.abs(b).max())
print("fallback rows", int(eng.stats[6].item()))

# ---- timings at the bench batch ----
Bb = 16384
xb = torch.from_numpy(synth.activations(Bb, D, seed=1, stream=0, bf16=True)).to(dev).to(torch.bfloat16)
h = eng.prepare(N.PREC_BF16, Bb, force=True)
lib = N.lib()
v = torch.empty(Bb, K, device=dev); i = torch.empty(Bb, K, dtype=torch.int32, device=dev); dp = torch.empty(Bb, K, device=dev)
stn = eng.stats.data_ptr(); s = eng.stream(); pk = eng.pack.data_ptr()
N.check(lib.wsae_profile_enable(h, -1, 64), "prof")
for it in range(12):
    N.check(lib.wsae_encode_topk(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, stn, s), "enc")
    N.check(lib.wsae_decode_loss(h, pk, xb.data_ptr(), N.DT_BF16, 0, v.data_ptr(), i.data_ptr(), Bb, 0, 1, dp.data_ptr(), 0, 0, stn, s), "dec")
torch.cuda.synchronize()
print("separate:", {k: round(ms / n * 1e3, 1) for k, (n, ms) in N.profile_read(h).items()})
N.check(lib.wsae_profile_disable(h), "x"); N.check(lib.wsae_profile_enable(h, -1, 64), "prof")
for it in range(12):
    N.check(lib.wsae_encode_decode(h, pk, xb.data_ptr(), N.DT_BF16, 0, Bb, v.data_ptr(), i.data_ptr(), 0, 0, 1, dp.data_ptr(), 0, stn, s), "encdec")
torch.cuda.synchronize()
print("fused   :", {k: round(ms / n * 1e3, 1) for k, (n, ms) in N.profile_read(h).items()})
print("fallback rows total", int(eng.stats[6].item()))
