"""What a library GEMM achieves on the dense layers' shapes (torch.mm -> hipBLASLt / rocBLAS), per network, B = 512."""
import torch
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
B, F, L2 = 512, 8192, 256
bf = torch.bfloat16
cases = {
  "head fwd  [B,F]x[F,2L]": (torch.randn(B, F, device="cuda", dtype=bf), torch.randn(F, L2, device="cuda", dtype=bf)),
  "head fwd  NT [B,F]x[2L,F]^T": (torch.randn(B, F, device="cuda", dtype=bf), torch.randn(L2, F, device="cuda", dtype=bf).t()),
  "d1 fwd    [B,256]x[256,F]": (torch.randn(B, 256, device="cuda", dtype=bf), torch.randn(256, F, device="cuda", dtype=bf)),
  "d1 dgrad  [B,F]x[F,256]": (torch.randn(B, F, device="cuda", dtype=bf), torch.randn(F, 256, device="cuda", dtype=bf)),
  "head dgrad [B,2L]x[2L,F]": (torch.randn(B, L2, device="cuda", dtype=bf), torch.randn(L2, F, device="cuda", dtype=bf)),
  "head wgrad [F,B]x[B,2L]": (torch.randn(B, F, device="cuda", dtype=bf).t(), torch.randn(B, L2, device="cuda", dtype=bf)),
  "d1 wgrad  [256,B]x[B,F]": (torch.randn(B, 256, device="cuda", dtype=bf).t(), torch.randn(B, F, device="cuda", dtype=bf)),
}
for k, (a, b) in cases.items():
    t = timeit(lambda: torch.mm(a, b))
    print("%-32s %7.1f us  %6.1f TF/s" % (k, t, 2.0 * a.shape[0] * a.shape[1] * b.shape[1] / t / 1e6))
# both networks as one batched call
a = torch.randn(2, B, F, device="cuda", dtype=bf); b = torch.randn(2, F, L2, device="cuda", dtype=bf)
print("head fwd bmm x2 %7.1f us" % timeit(lambda: torch.bmm(a, b)))
a = torch.randn(2, B, F, device="cuda", dtype=bf).transpose(1, 2); b = torch.randn(2, B, L2, device="cuda", dtype=bf)
print("head wgrad bmm x2 %7.1f us" % timeit(lambda: torch.bmm(a, b)))
