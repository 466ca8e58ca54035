"""The two launches tools/profile_pmc_igemm.sh counts: the Winograd batched GEMM of the 16x16 256->256 layer at the conditioning
batch (B = 576) and the direct 3x3 conv of the 64x64 64->64 layer at B = 64, five launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvg_amd import ops
from dvg_amd._lib import lib
dev = torch.device("cuda:0"); p = ops._p; s = ops._stream
N, H, C, Cout = 576, 16, 256, 256
T = N * (H // 4) ** 2
v = torch.randn((36, T, C), device=dev); m = torch.empty((36, T, Cout), device=dev)
u = ops.winograd_weight(torch.randn(Cout, C, 3, 3, device=dev) * 0.02, 4)
for _ in range(5):
    lib().dvg_gemm_batched_k16(p(v), p(u), p(m), 36, T // 16, 16, C, Cout, s())
x = ops.nhwc_empty(64, 64, 64, 64, dev).normal_()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.02
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
wp = ops.pack_igemm_weight(w)
for _ in range(5):
    ops.conv3x3(x, None, wp, sc, sh)
torch.cuda.synchronize()
