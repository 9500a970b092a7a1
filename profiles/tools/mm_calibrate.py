"""Calibration only: what the vendor GEMM (hipBLASLt through torch.matmul) reaches on the two contraction shapes
of the train step.  Not part of the product path or of bench.py."""
import torch, time
dev='cuda'
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1000
B,D,H=16384,384,3072
x=torch.randn(B,D,device=dev,dtype=torch.bfloat16); W=torch.randn(H,D,device=dev,dtype=torch.bfloat16)
out=torch.empty(B,H,device=dev,dtype=torch.bfloat16)
t=bench(lambda: torch.matmul(x,W.t(),out=out)); print("encode NT bf16 out: %.1f us  %.0f TF"%(t, 2*B*D*H/t/1e6))
out32=torch.empty(B,H,device=dev,dtype=torch.float32)
hT=torch.randn(H,B,device=dev,dtype=torch.bfloat16); g=torch.randn(B,D,device=dev,dtype=torch.bfloat16)
o2=torch.empty(H,D,device=dev,dtype=torch.bfloat16)
t=bench(lambda: torch.matmul(hT,g,out=o2)); print("wgrad [H,B]x[B,D]: %.1f us  %.0f TF"%(t, 2*B*D*H/t/1e6))
gT=g.t().contiguous()
t=bench(lambda: torch.matmul(hT,gT.t(),out=o2)); print("wgrad NT [H,B]x[D,B]^T: %.1f us  %.0f TF"%(t, 2*B*D*H/t/1e6))
a=torch.randn(8192,8192,device=dev,dtype=torch.bfloat16); b=torch.randn(8192,8192,device=dev,dtype=torch.bfloat16)
t=bench(lambda: torch.matmul(a,b.t()),10); print("8192^3 NT: %.1f us %.0f TF"%(t, 2*8192**3/t/1e6))
