"""Static guard on the compiled MFMA contraction kernels (CPU-side: hipcc cross-compiles k_gemm.hip to gfx950 assembly): no basic block
that issues MFMAs may touch scratch.  Round 5 lost a factor two on k_tgemm to a register spill the compiler placed INSIDE the K loop of
the partial-tile variant after an epilogue edit - invisible in the parity tests, visible only in the timing."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("unit,kernels", [("k_gemm", ("k_tgemm", "k_syrk")), ("k_small", ("k_small_tail",)),
                                          ("k_syrk_small", ("k_syrk_small",))])
def test_no_scratch_traffic_inside_the_mfma_loops(tmp_path, unit, kernels):
    asm = tmp_path / (unit + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "gpz_amd", "csrc"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "gpz_amd", "csrc", unit + ".hip"), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    lines = asm.read_text().splitlines()
    kernel = block = None
    nmfma = nscratch = 0
    bad, seen = [], set()

    def close():
        if kernel and nmfma and nscratch:
            bad.append((kernel, block, nmfma, nscratch))

    for l in lines:
        m = re.match(r"^(_Z\w*(%s)\w*):" % "|".join(kernels), l)
        if m:
            close()
            kernel, block, nmfma, nscratch = m.group(1), "entry", 0, 0
            seen.add(m.group(2))
            continue
        if kernel is None:
            continue
        if l.startswith(".Lfunc_end"):
            close()
            kernel = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            close()
            block, nmfma, nscratch = m.group(1), 0, 0
            continue
        t = l.strip()
        if t.startswith("v_mfma"):
            nmfma += 1
        elif t.startswith("scratch_"):
            nscratch += 1
    assert seen == set(kernels), seen
    assert not bad, bad
