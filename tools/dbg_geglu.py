import math, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Ctx, ptr
from tests.test_gpu_kernels import ilv32, h16
ctx = Ctx(); DEV="cuda"
M,K,I=64,64,64
a=h16(M,K,seed=51); w=h16(2*I,K,scale=1/math.sqrt(K),seed=52); bias=torch.zeros(2*I,device=DEV)
wi=ilv32(w[:I].t(), w[I:].t()).t().contiguous()
out=torch.zeros(M,I,dtype=torch.half,device=DEV)
ctx.call("pnpi_op_gemm_geglu", ptr(a),K,ptr(wi),K,M,2*I,K,ptr(bias),ptr(out),I); torch.cuda.synchronize()
h=a.float()@w.float().t()
ref=h[:,:I]*F.gelu(h[:,I:])
print("out[0,:8]",out[0,:8].float().cpu()); print("ref[0,:8]",ref[0,:8].cpu())
# plain gemm of interleaved weights for comparison
o2=torch.zeros(M,2*I,dtype=torch.half,device=DEV)
ctx.call("pnpi_op_gemm", ptr(a),K,ptr(wi),K,M,2*I,K,1.0,None,None,ptr(o2),2*I,1<<30,None,0,0,1,-1,0); torch.cuda.synchronize()
hi=a.float()@wi.float().t()
print("plain packed gemm err", ((o2.float()-hi).norm()/hi.norm()).item())
x=o2.float(); 
cand=x[:, :32]*F.gelu(x[:,32:64])
print("cand0 vs out[:, :32]", ((out[:,:32].float()-cand).norm()/cand.norm()).item(), "ref vs cand", ((ref[:,:32]-cand).norm()/cand.norm()).item())
print("h pairing check: packed col0..3", hi[0,:4].cpu(), "h x cols", h[0,:4].cpu(), "packed 32..35", hi[0,32:36].cpu(), "gate", h[0,I:I+4].cpu())
