"""Per layer pair, the two forms of the x half of a decoder upsample + concat conv with the skip half hoisted (eval rollouts):
the 4-tap transposed conv (K4 form) against Winograd F(4x4) over the upsampled map (`dvg_winograd_input(upsample=1)` + addend in
the output / hand-over kernel), each followed by the block's second 3x3 layer.  Shapes: vgg_64 decoder at B = 64.
    python tools/bench_upconv_form.py"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops
from tools.bench_small import time_fn
dev=torch.device('cuda:0')
for (N,H,C1,Cout,Cn) in [(64,4,512,512,512),(64,8,256,256,256),(64,16,128,128,128)]:
    x=ops.nhwc_empty(N,C1,H,H,dev).normal_()
    w=torch.randn(Cout,C1,3,3,device=dev)*0.02; w2=torch.randn(Cn,Cout,3,3,device=dev)*0.02
    sc,sh=torch.rand(Cout,device=dev)+0.5, torch.randn(Cout,device=dev)*0.1
    S=ops.nhwc_empty(N,Cout,2*H,2*H,dev).normal_()
    u=ops.winograd_weight(w,4); u2=ops.winograd_weight(w2,4)
    k4=torch.zeros((Cout,C1,4,4),device=dev)
    for ty in range(3):
        for tx in range(3):
            k4[:,:,2-ty:4-ty,2-tx:4-tx]+=w[:,:,ty:ty+1,tx:tx+1]
    kp=ops.pack_igemm_weight(k4.permute(1,0,2,3).contiguous(),transposed=True)
    chain = ops.winograd_chain_ok(N,Cout,2*H,2*H)
    def old():
        y=ops.convT4x4s2(x,None,kp,sc,sh,addend=S)
        return ops.conv3x3_winograd(y,u2,sc,sh)
    def new():
        v=ops.conv3x3_winograd(x,u,sc,sh,upsample=True,addend=S,to_v=chain)
        return ops.conv3x3_winograd(v,u2,sc,sh)
    t_old=time_fn(old, iters=100); t_new=time_fn(new, iters=100)
    t_ct=time_fn(lambda: ops.convT4x4s2(x,None,kp,sc,sh,addend=S), iters=100)
    t_w=time_fn(lambda: ops.conv3x3_winograd(x,u,sc,sh,upsample=True,addend=S,to_v=chain), iters=100)
    print(f"x {C1}@{H}^2 -> {Cout}@{2*H}^2 (+next {Cn}): convT+next {t_old:6.1f} us | wino+next {t_new:6.1f} us | layer alone: convT {t_ct:6.1f} wino(to_v={chain}) {t_w:6.1f}")
