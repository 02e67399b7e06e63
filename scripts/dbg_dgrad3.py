import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import torch_ref
from split_vae_amd import ops
g = torch.Generator().manual_seed(0)
B, H, Cin, Cout, k, s = 4, 32, 3, 32, 3, 2
w = torch.randn(k, k, Cin, Cout, generator=g) * 0.1
xr = torch.randn(B, H, H, Cin, generator=g).double().requires_grad_(True)
yr = torch_ref.conv2d_same(xr, w.double(), None, s, None)
gy = torch.randn(yr.shape, generator=g)
yr.backward(gy.double())
c = ops.Conv2D(B, H, H, Cin, Cout, k, s, dtype=torch.float32)
c.prep(w.cuda())
dx = c.dgrad(gy.cuda().contiguous(), None).cpu()
print(dx.shape)
ref = xr.grad
for ch in range(8):
    a = dx[..., ch].double()
    if ch < Cin:
        print(ch, float((a - ref[..., ch]).norm() / ref[..., ch].norm()), [float((a - ref[..., c2]).norm() / ref[..., c2].norm()) for c2 in range(Cin)])
    else:
        print(ch, "pad abs max", float(a.abs().max()))
print(dx[0, 5, 5], ref[0, 5, 5])
print(dx[0, 5, 6], ref[0, 5, 6])
