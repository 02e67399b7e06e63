"""Build libsplitvae_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, os.environ.get("SV_LIB_NAME", "libsplitvae_hip.so"))
OBJ_TAG = os.environ.get("SV_OBJ_TAG", "")          # build variants side by side (kernel A/B experiments)
EXTRA = os.environ.get("SV_EXTRA_FLAGS", "").split()
SOURCES = ["pointwise.hip", "gm_pointwise.hip", "tap_gemm.hip", "tile_conv.hip", "row_conv.hip", "poly_fix.hip", "poly_wgrad.hip", "polyc_wgrad.hip", "polyd_dgrad.hip", "stn.hip", "spair_render.hip", "spair_loss.hip", "wgrad.hip", "wgrad_tile.hip", "wgrad_tile_f32.hip", "wgrad_roll.hip", "wgrad_p5.hip", "wgrad_e1.hip", "wgrad_e2.hip", "conv_api.hip", "lgvae_plan.hip", "gm_encoder.hip", "hostio.hip", "comm.hip", "dense_f32.hip", "tape.hip", "latent_gemm.hip", "streams.hip"]
# per-file extra flags (measured and rejected for row_conv.hip: -fno-slp-vectorize with a scalar fp32 blend -- +7 % time, the
# packed v_pk_fma_f32 blend issues half the instructions)
FILE_FLAGS = {}
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=off"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "splitvae.h"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", OBJ_TAG + ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + FILE_FLAGS.get(s, []) + EXTRA + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(len(jobs), 6) or 1) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
