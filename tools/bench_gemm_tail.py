"""Batched GEMM (igemm GEMM mode) time against the number of positions NB at the vgg_64 Winograd shapes: how much of a launch is
the partially filled last residency round.  GPU only."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops
from dvg_amd._lib import lib
from tools.bench_small import time_fn
dev=torch.device('cuda:0'); p=ops._p; s=ops._stream
for (T,C,Cout) in [(256,512,512),(1024,256,256),(4096,128,128),(256,512,256)]:
    for NB in (28,32,36,40,48,64):
        v=torch.randn((NB,T,C),device=dev); m=torch.empty((NB,T,Cout),device=dev)
        u=torch.randn((NB,C//16,1,Cout,ops.packed_row_floats()),device=dev)*0.02
        t=time_fn(lambda: lib().dvg_gemm_batched_k16(p(v),p(u),p(m),NB,T//16,16,C,Cout,s()))
        wgs=NB*(T//64)*(Cout//64)
        print(f"T={T} {C}->{Cout} NB={NB:2d} wgs={wgs:5d} ({wgs/1024:.3f} rounds) {t:7.1f} us  {2e-6*NB*T*C*Cout/t:6.1f} TF")
