"""Timeline of one workgroup of wgrad_roll_kernel from its in-kernel clock stamps (a -DSV_ROLL_STAMP build of the library:
SV_EXTRA_FLAGS=-DSV_ROLL_STAMP SV_OBJ_TAG=_stamp SV_LIB_NAME=libsplitvae_stamp.so python split_vae_amd/build.py; run with
SV_LIB_NAME=libsplitvae_stamp.so).  Prints, per step and wave, the shader-clock offsets of the stamp points."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from split_vae_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
conv = ops.Conv2D(B, 32, 32, 64, 32, 6, 1, act="relu", dtype=torch.bfloat16, ups_in=True)
conv.prep(torch.randn(6, 6, 64, 32, device="cuda") * 0.05)
x = torch.randn(B, 16, 16, 64, device="cuda").bfloat16()
dy = torch.randn(B, 32, 32, 32, device="cuda").bfloat16()
dw = torch.zeros(6, 6, 64, 32, device="cuda"); db = torch.zeros(32, device="cuda")
lib = ops._lib.load()
n = lib.sv_conv2d_wgrad_workspace_bytes(C.byref(conv.desc))
ws = torch.zeros((n,), dtype=torch.uint8, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    lib.sv_conv2d_nhwc_wgrad_ws(C.byref(conv.desc), P(x), P(dy), P(dw), P(db), P(ws), C.c_int64(n), C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
st = ws[:4 * 8 * 8 * 8].view(torch.int64).cpu().view(4, 8, 8)
t0 = int(st[0, :, 0].min())
names = ["top", "staged", "mfma0", "mfma3", "mfma_end", "waited", "barrier_out", "-"]
for step in range(4):
    print("step", step)
    for w in range(8):
        print("  wave %d: " % w + "  ".join("%s %6d" % (names[i], int(st[step, w, i]) - t0) for i in range(7)))
