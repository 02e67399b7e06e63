"""Per-layer micro-benchmark (GPU): forward / dgrad / wgrad of every conv layer of the model at the
bench batch size, through the public C ABI.  Usage: python scripts/bench_layers.py [B] [layers...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from split_vae_amd import ops

LAYERS = {  # name: H, Cin, Cout, k, stride, act, y_f32
    "e1": (64, 3, 32, 6, 2, "relu", False),
    "e2": (32, 32, 64, 6, 2, "relu", False),
    "e3": (16, 64, 128, 4, 2, "relu", False),
    "d2": (8, 128, 128, 4, 1, "relu", False),
    "d3": (16, 128, 64, 4, 1, "relu", False),
    "d4": (32, 64, 32, 6, 1, "relu", False),
    "d5": (64, 32, 6, 6, 1, None, True),
    "e1s2d": (32, 16, 32, 3, 1, "relu", False),      # e1 on the space-to-depth input (2x2 pixels x 3 channels -> 12 of 16): 3x3 stride 1
}


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    names = sys.argv[2:] or list(LAYERS)
    which = os.environ.get("SV_BENCH_OPS", "fwd,dgrad,wgrad").split(",")
    for name in names:
        H, Cin, Cout, k, s, act, yf32 = LAYERS[name]
        ups = name in ("d3", "d4", "d5") and not os.environ.get("SV_NO_FUSED_UPSAMPLE")   # as the training plan runs them
        conv = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=act, dtype=torch.bfloat16, y_f32=yf32, ups_in=ups)
        w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
        conv.prep(w)
        HI = H // 2 if ups else H
        x = torch.randn(B, HI, HI, conv.desc.ldx, device="cuda").to(torch.bfloat16)
        bias = torch.zeros(Cout, device="cuda")
        OH = conv.OH
        dy = torch.randn(B, OH, OH, (Cout + 7) // 8 * 8, device="cuda").to(torch.bfloat16)
        flops = 2.0 * B * OH * OH * Cout * k * k * Cin
        dw = torch.zeros(k, k, Cin, Cout, device="cuda")
        db = torch.zeros(Cout, device="cuda")
        dx = torch.zeros(B, H, H, conv.desc.ldx, device="cuda", dtype=torch.bfloat16)
        y = conv.fwd(x, bias)
        import ctypes as C
        lib = ops._lib.load()
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())
        res = []
        if "fwd" in which:
            nws = lib.sv_conv2d_fwd_workspace_bytes(C.byref(conv.desc))
            fws = torch.empty((max(nws, 16),), dtype=torch.uint8, device="cuda")
            t = timeit(lambda: lib.sv_conv2d_nhwc_fwd_ws(C.byref(conv.desc), P(x), P(conv.w_fwd), P(bias), P(y), P(fws), nws, st()))
            res.append("fwd %7.1f us %6.0f TF/s" % (t, flops / t / 1e6))
        if "dgrad" in which and name != "e1":
            t = timeit(lambda: lib.sv_conv2d_nhwc_dgrad(C.byref(conv.desc), P(dy), P(conv.w_dgrad), None, P(dx), 0, st()))
            res.append("dgrad %7.1f us %6.0f TF/s" % (t, flops / t / 1e6))
        if "wgrad" in which:
            n = lib.sv_conv2d_wgrad_workspace_bytes(C.byref(conv.desc))
            ws = torch.empty((n,), dtype=torch.uint8, device="cuda")
            pdb = None if os.environ.get("SV_BENCH_NOBIAS") else P(db)        # A/B: the bias gradient's share of the launch
            t = timeit(lambda: lib.sv_conv2d_nhwc_wgrad_ws(C.byref(conv.desc), P(x), P(dy), P(dw), pdb, P(ws), C.c_int64(n), st()))
            res.append("wgrad %7.1f us %6.0f TF/s" % (t, flops / t / 1e6))
        print("%-3s B=%d  %s" % (name, B, "  |  ".join(res)), flush=True)


if __name__ == "__main__":
    main()
