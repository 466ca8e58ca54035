import sys, torch
one=torch.load(sys.argv[1]); two=torch.load(sys.argv[2])
a,b=two["grads_first_backward"],one["grads_first_backward"]
for k in b:
    den=float(b[k].abs().max())
    if den==0: continue
    d=float((a[k]-b[k]).abs().max())/den
    l2=float((a[k]-b[k]).double().norm()/b[k].double().norm())
    print(f"{k:50s} max-norm {d:.2e} rel-L2 {l2:.2e} |g|max {den:.2e}")
