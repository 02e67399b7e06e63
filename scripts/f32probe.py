"""fp32 weight gradient per layer (twin-sized launch: 1024 images), tile kernel with workspace vs the im2col kernel."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from split_vae_amd import ops
LAYERS = {"e1": (64, 3, 32, 6, 2), "e2": (32, 32, 64, 6, 2), "e3": (16, 64, 128, 4, 2), "d2": (8, 128, 128, 4, 1), "d3": (16, 128, 64, 4, 1), "d4": (32, 64, 32, 6, 1), "d5": (64, 32, 6, 6, 1)}
B = 1024
names = sys.argv[1:] or list(LAYERS)
out = []
for name in names:
    H, Cin, Cout, k, s = LAYERS[name]
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=None, dtype=torch.float32, y_f32=True)
    x = torch.randn(B, H, H, conv.desc.ldx, device="cuda")
    OH = H // s
    dy = torch.randn(B, OH, OH, (Cout + 7) // 8 * 8, device="cuda")
    dw = torch.zeros(k, k, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
    fl = 2.0 * B * OH * OH * k * k * Cin * Cout
    def t(ws):
        for _ in range(2): conv.wgrad(x, dy, workspace=ws, dw=dw, db=db)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): conv.wgrad(x, dy, workspace=ws, dw=dw, db=db)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 5
    tt = t(True)
    out.append("%s %.3f ms %.0f TF" % (name, tt, fl / tt / 1e9))
print("  ".join(out))
