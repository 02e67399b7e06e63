"""Is the fp32 step run-to-run identical on the DEFAULT path (no SV_DETERMINISTIC)?  sha256 of five training steps (losses, gradients, weights), each configuration twice
in one process and once more in a fresh process.  Since round 5 the fp32 latent block runs on latent_gemm.hip (slabs summed in slice order, whole-batch weight-gradient tiles):
no fp32 atomics are left in the default fp32 SPLIT-VAE step."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r03_step_hash import run

CASES = (("f32 64x64 B=64", ("f32", 64, 64, 8)), ("f32 64x64 B=512", ("f32", 64, 512, 8)), ("f32 32x32 B=64", ("f32", 32, 64, 4)))
if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name, a in CASES:
            print(name, run(*a)[:24], flush=True)
        sys.exit(0)
    for name, a in CASES:
        print(name, run(*a)[:24], run(*a)[:24], flush=True)
    print("fresh process:")
    print(subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True).stdout)
