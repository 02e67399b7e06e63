"""fp32 conv op (fwd / dgrad / wgrad / dbias) on the SPAIR geometries vs torch fp64 autograd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import torch_ref
from split_vae_amd import torch_ops as T
g = torch.Generator().manual_seed(0)
CASES = [(48, 32, 3, 32, 3, 2, False), (48, 16, 32, 64, 3, 2, False), (3, 48, 3, 128, 4, 2, False), (3, 24, 128, 128, 4, 2, False),
                                     (3, 12, 128, 128, 4, 3, False), (3, 4, 128, 128, 1, 1, False), (3, 4, 128, 100, 1, 1, False),
                                     (48, 8, 32, 64, 3, 1, False), (48, 8, 64, 32, 3, 1, True), (48, 16, 32, 4, 3, 1, True),
                                     (3, 6, 128, 128, 3, 1, False), (3, 6, 128, 64, 3, 1, True), (3, 12, 64, 32, 3, 1, True), (3, 24, 32, 3, 3, 1, True)]
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (B, H, Cin, Cout, k, s, ups) in CASES:
    x = torch.randn(B, H, H, Cin, generator=g)
    w = torch.randn(k, k, Cin, Cout, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    xin = torch_ref.resize_bilinear_2x(xr) if ups else xr
    yr = torch_ref.conv2d_same(xin, wr, br, s, "relu")
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    r8 = lambda v: (v + 7) // 8 * 8
    xd = torch.nn.functional.pad(x, (0, r8(Cin) - Cin)).cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    try:
        yd = T.conv2d(xd, wd, bd, s, "relu", ups, False)
        yd[..., :Cout].backward(gy.cuda())
    except Exception as e:
        print((B, H, Cin, Cout, k, s, ups), "FAILED", str(e)[:80])
        continue
    rel = lambda a, c: float((a.double().cpu() - c).norm() / c.norm())
    print((B, H, Cin, Cout, k, s, ups), "y %.1e dx %.1e dw %.1e db %.1e" % (rel(yd[..., :Cout], yr.detach()), rel(xd.grad[..., :Cin], xr.grad), rel(wd.grad, wr.grad), rel(bd.grad, br.grad)))
