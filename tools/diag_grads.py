import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dvg_oracle as orc, params
from tests.common import backbone_case, rel_err
tag, family = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
enc, dec, esd, dsd, x, vec = backbone_case(tag)
gy = params.normal(900, *x.shape); gh = params.normal(901, x.shape[0], 90)
mk = lambda sd: {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
e, d = mk(esd), mk(dsd)
dbl = len(sys.argv) > 3
if dbl:
    e = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v)) for k, v in e.items()}
    d = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v)) for k, v in d.items()}
    x64 = x.double()
else:
    x64 = x
if family == "vgg":
    h, skips = orc.vgg_encoder(x64, e, True); y = orc.vgg_decoder(h, skips, d, True)
else:
    h, skips = orc.dcgan_encoder(x64, e, True); y = orc.dcgan_decoder(h, skips, d, True, "tanh")
((y * gy.to(y.dtype)).sum() + (h * gh.to(y.dtype)).sum()).backward()
enc.to(dev).train(); dec.to(dev).train()
ho, so = enc(x.to(dev)); yo = dec([ho, so])
((yo * gy.to(dev)).sum() + (ho * gh.to(dev)).sum()).backward()
print("fwd err", rel_err(yo, y), rel_err(ho, h))
for name, ref, ours in (("enc", e, dict(enc.named_parameters())), ("dec", d, dict(dec.named_parameters()))):
    for k, p in ours.items():
        g = ref[k].grad
        scale = float(g.abs().max())
        err = float((p.grad.double().cpu() - g.double()).abs().max()) / max(scale, 1e-12)
        print(f"{name} {k:28s} scale {scale:10.4g} relerr {err:.2e}")
