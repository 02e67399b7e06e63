"""Surrogate timing for the polyphase form of the decoder's upsample+conv layers: a plain 5x5 stride-1 conv on the LOW-RES grid
with 4*Cout output columns (d5: 32 -> 32 on 32x32; d4: 64 -> 128 on 16x16), through the public conv ABI."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

B = 1024
for (H, Cin, Cout, k) in ((32, 32, 32, 5), (16, 64, 128, 5), (32, 32, 32, 6), (8, 128, 256, 4)):
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.bfloat16)
    conv.prep(torch.randn(k, k, Cin, Cout, device="cuda") * 0.05)
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    bias = torch.zeros(Cout, device="cuda")
    t = timeit(lambda: conv.fwd(x, bias))
    fl = 2.0 * B * H * H * Cout * k * k * Cin
    print("conv %dx%d k%d %d->%d B=%d: %.1f us  %.0f TF/s" % (H, H, k, Cin, Cout, B, t, fl / t / 1e6))
