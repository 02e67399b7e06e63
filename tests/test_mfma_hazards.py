"""CPU: no kernel of the library reads an MFMA result before the ISA's wait states have passed (scripts/mfma_hazard_scan.py).

Round 5 found `ds_write_b128 v, a[0:3]` two instructions behind the last `v_mfma_f32_16x16x4_f32 a[0:3]` of the previous basic block in polyd_edge_kernel
(polyd_dgrad.hip): hipcc 7.2 pads MFMA -> VALU / LDS / VMEM reads inside a block but did not carry the count across that s_branch.  The build that shipped
happened to work; the same source with one more kernel argument returned a stale third accumulator component, differently from run to run.  The kernel now spells the
wait states out; this test keeps every kernel of every .hip file honest: all sources are compiled to gfx950 assembly (no GPU needed) and scanned along the control flow."""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc (cross-compiles gfx950 assembly without a GPU)")
def test_no_mfma_result_is_read_early(tmp_path):
    srcs = sorted(glob.glob(os.path.join(ROOT, "split_vae_amd", "csrc", "*.hip")))
    assert len(srcs) >= 20
    hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")

    def asm(src):
        out = str(tmp_path / (os.path.basename(src)[:-4] + ".s"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-everything", "-S", "--cuda-device-only", "-o", out, src],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        outs = list(ex.map(asm, srcs))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "mfma_hazard_scan.py")] + outs, capture_output=True, text=True)
    found = [l for l in r.stdout.splitlines() if "reads the result of the MFMA" in l]
    assert r.returncode == 0 and not found, "\n".join(found[:20]) + r.stderr[-500:]
