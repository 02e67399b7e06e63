import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import ops
B = 128
def mk(H, Cin, Cout, k, s, ups=False):
    c = ops.Conv2D(B, H, H, Cin, Cout, k, s, act="relu", dtype=torch.bfloat16, ups_in=ups)
    c.prep(torch.randn(k, k, Cin, Cout, device="cuda") * 0.05)
    return c
e1 = mk(64, 3, 32, 6, 2)
d4 = mk(32, 64, 32, 6, 1, ups=True)
d3 = mk(16, 128, 64, 4, 1, ups=True)
x = torch.randn(B, 64, 64, 8, device="cuda").bfloat16(); dy1 = torch.randn(B, 32, 32, 32, device="cuda").bfloat16()
dy4 = torch.randn(B, 32, 32, 32, device="cuda").bfloat16(); dy3 = torch.randn(B, 16, 16, 64, device="cuda").bfloat16()
dw = torch.zeros(6, 6, 3, 32, device="cuda"); db = torch.zeros(32, device="cuda")
def t_e1(pre):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(20):
        pre()
        a.record(); e1.wgrad(x, dy1, workspace=True, dw=dw, db=db); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / 20 * 1e3
print("e1 wgrad alone            %.1f us" % t_e1(lambda: None))
print("after d4 dgrad (row/tile) %.1f us" % t_e1(lambda: d4.dgrad(dy4)))
print("after d3 dgrad            %.1f us" % t_e1(lambda: d3.dgrad(dy3)))
print("e1 wgrad alone again      %.1f us" % t_e1(lambda: None))
