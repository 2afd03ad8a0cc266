"""Numerical feasibility of an fp16 hi/lo split (3 products) for the head chain, emulated on the CPU."""
import numpy as np, torch
torch.manual_seed(0)
rows = 2048
dims = [259, 256, 256, 512, 512, 1024]
x = torch.randn(rows, 259)
x[:, :3] *= 20          # xyz in metres
x[:, 3:] = torch.relu(x[:, 3:])
ws = [torch.empty(b, a).uniform_(-1, 1) * (6.0 / (a + b)) ** 0.5 for a, b in zip(dims[:-1], dims[1:])]
bs = [torch.zeros(b) for b in dims[1:]]

def split(t, scale=2048.0):
    hi = t.half().float()
    lo = ((t - hi) * scale).half().float()
    return hi, lo, scale

def mm_split(a, w):
    ah, al, s = split(a); wh, wl, _ = split(w)
    main = ah @ wh.t()
    corr = ah @ wl.t() + al @ wh.t()
    return main + corr / s

def mm_bf16x3(a, w):
    def sp(t):
        p0 = t.bfloat16().float(); r = t - p0; p1 = r.bfloat16().float(); p2 = (r - p1).bfloat16().float(); return p0, p1, p2
    a0, a1, a2 = sp(a); w0, w1, w2 = sp(w)
    return a0 @ w0.t() + (a0 @ w1.t() + a1 @ w0.t()) + (a0 @ w2.t() + a1 @ w1.t() + a2 @ w0.t())

def chain(mm, dt=torch.float32):
    h = x.to(dt)
    for w, b in zip(ws, bs):
        h = torch.relu(mm(h, w.to(dt)) + b.to(dt))
    return h.max(dim=0).values

ref = chain(lambda a, w: a @ w.t(), torch.float64)
for name, mm in (('fp32', lambda a, w: a @ w.t()), ('fp16 hi/lo 3 products', mm_split), ('bf16 x3, 6 products', mm_bf16x3),
                 ('fp16 plain', lambda a, w: a.half().float() @ w.half().float().t())):
    out = chain(mm).double()
    err = (out - ref).abs()
    print('{:24s} max abs {:.3e}  max rel {:.3e}  mean rel {:.3e}'.format(name, err.max().item(), (err / ref.abs().clamp(min=1e-6)).max().item(), (err / ref.abs().clamp(min=1e-6)).mean().item()))
