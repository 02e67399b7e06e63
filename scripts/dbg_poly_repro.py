import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda").manual_seed(7)
conv = ops.Conv2D(B, 64, 64, 32, 6, 6, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
w = (torch.rand(6, 6, 32, 6, device="cuda", generator=g) * 2 - 1) * 0.1
conv.prep(w)
x = torch.randn(B, 32, 32, 32, device="cuda", generator=g).bfloat16()
bias = torch.randn(6, device="cuda", generator=g) * 0.1
ys = [conv.fwd(x, bias).clone() for _ in range(4)]
for i in range(1, 4):
    d = (ys[i] != ys[0])
    print("run", i, "mismatches", int(d.sum()))
    if d.any():
        idx = d.nonzero()[:10]
        print(idx.tolist())
        print((ys[i] - ys[0])[d][:10].tolist())
        bs = d.any(dim=3).any(dim=2).any(dim=1).nonzero().flatten()
        print("images", bs[:20].tolist(), "rows", d.any(dim=3).any(dim=2).any(dim=0).nonzero().flatten().tolist(), "cols", d.any(dim=3).any(dim=1).any(dim=0).nonzero().flatten().tolist())
# is the workspace itself reproducible?
ws = []
for _ in range(3):
    conv.fwd(x, bias); torch.cuda.synchronize()
    ws.append(conv._fws.clone().view(torch.float32))
used = (5 * 64 + 6 * 64) * 6
per = ws[0].numel() // B
print("per-image floats (allocated)", per, "used", used)
for i in (1, 2):
    d = (ws[i] != ws[0]) & ~(torch.isnan(ws[i]) & torch.isnan(ws[0]))
    idx = d.nonzero().flatten()
    print("ws run", i, "mismatch", idx.numel(), "images", sorted(set((idx // per).tolist()))[:10], "offsets", sorted(set((idx % per).tolist()))[:12], sorted(set((idx % per).tolist()))[-5:])
d = (ws[1] != ws[0])
idx = d.nonzero().flatten()
if idx.numel():
    used_img = (idx // used)
    print("by used stride: images", sorted(set(used_img.tolist()))[:10], "local offsets", sorted(set((idx % used).tolist()))[:8], "...", sorted(set((idx % used).tolist()))[-4:])
    j = idx[:12]
    for k in range(3):
        print("run", k, [round(float(v), 5) for v in ws[k][j]])
    # neighbours before the first mismatch
    j0 = int(idx[0])
    print("around first mismatch", j0, [round(float(v), 5) for v in ws[0][j0 - 6:j0 + 12]], [round(float(v), 5) for v in ws[1][j0 - 6:j0 + 12]])
