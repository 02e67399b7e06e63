"""EXPERIMENT: the whole polyphase weight gradient of d5 vs the fp64 reference and the current kernel; timing."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import ops
from oracle import torch_ref
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = ops._lib.load()
conv = ops.Conv2D(B, H, H, 32, 6, 6, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, H // 2, H // 2, 32, device="cuda", generator=g).bfloat16()
dy = torch.randn(B, H, H, 8, device="cuda", generator=g).bfloat16(); dy[..., 6:] = 0
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.sv_dbg_poly_wgrad_ws_bytes.restype = C.c_int64
nws = lib.sv_dbg_poly_wgrad_ws_bytes(C.byref(conv.desc))
ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
dw = torch.zeros(6, 6, 32, 6, device="cuda"); db = torch.zeros(6, device="cuda")
fn = lib.sv_dbg_poly_wgrad; fn.restype = C.c_int
rc = fn(C.byref(conv.desc), P(x), P(dy), P(dw), P(db), P(ws), st); torch.cuda.synchronize(); print("rc", rc)
dw0 = torch.zeros_like(dw); db0 = torch.zeros_like(db)
conv.wgrad(x, dy, workspace=True, dw=dw0, db=db0); torch.cuda.synchronize()
if B <= 64:
    wt = torch.zeros(6, 6, 32, 6, dtype=torch.float64, requires_grad=True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(x.double().cpu()), wt, torch.zeros(6, dtype=torch.float64), 1, None)
    (y * dy[..., :6].double().cpu()).sum().backward()
    ref = wt.grad.cuda()
    print("poly vs fp64:    %.2e   current vs fp64: %.2e" % (float((dw.double() - ref).norm() / ref.norm()), float((dw0.double() - ref).norm() / ref.norm())))
    bad = (dw.double() - ref).abs()
    print("max abs err by (ky,kx):", bad.amax(dim=(2, 3)).cpu().numpy().round(4).tolist())
print("poly vs current: %.2e  bias %.2e" % (float((dw - dw0).norm() / dw0.norm()), float((db - db0).abs().max() / db0.abs().max())))
dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
fn(C.byref(conv.desc), P(x), P(dy), P(dw2), P(db2), P(ws), st); torch.cuda.synchronize()
print("second call equal:", bool(torch.equal(dw, dw2)), bool(torch.equal(db, db2)))
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
print("poly wgrad total: %.1f us   current: %.1f us" % (timeit(lambda: fn(C.byref(conv.desc), P(x), P(dy), P(dw), P(db), P(ws), st)),
                                                         timeit(lambda: conv.wgrad(x, dy, workspace=True, dw=dw0, db=db0))))
