"""The polyphase weight gradient of the decoder head (sv_conv2d_nhwc_wgrad_poly: main term on wgrad_tile_kernel<7,...> + slab reduce +
frame kernel + its reduce + projection) at 2 x 512 images, stand-alone; with a debug-knob build SV_WT_DBG ablates the main term's
phases (1 skip flush(+reduce), 2 skip input staging, 4 skip dY staging, 8 skip the MFMA loop).  usage: python scripts/r03_abl_polywgrad.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H, Cin, Cout, k = 64, 32, 6, 6
conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
conv.prep(torch.zeros(k, k, Cin, Cout).cuda())
x = torch.randn(B, H // 2, H // 2, Cin, device="cuda").bfloat16()
dy = torch.randn(B, H, H, 8, device="cuda").bfloat16()
dy[..., Cout:] = 0
for _ in range(5):
    conv.wgrad_poly(x, dy)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    conv.wgrad_poly(x, dy)
torch.cuda.synchronize()
print("wgrad_poly d5 B=%d  %.1f us" % (B, (time.perf_counter() - t0) / 50 * 1e6))
