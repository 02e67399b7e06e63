"""e2's input gradient (merged parity classes, masked) alone: time per launch; with a -DSV_DEBUG_KNOBS build and SV_RC_STAMP=1 the row-ring kernel prints its phase shares."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import ops
B = 1024
c = ops.Conv2D(B, 32, 32, 32, 64, 6, 2, act="relu", dtype=torch.bfloat16); c.prep(torch.randn(6, 6, 32, 64, device="cuda") * 0.05)
dy = torch.randn(B, 16, 16, 64, device="cuda").bfloat16(); mask = torch.randn(B, 32, 32, 32, device="cuda").bfloat16()
for _ in range(20):
    dx = c.dgrad(dy, relu_mask=mask)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    dx = c.dgrad(dy, relu_mask=mask)
b.record(); torch.cuda.synchronize()
print("dgrad.e2 B=%d: %.1f us" % (B, a.elapsed_time(b) / 20 * 1e3))
