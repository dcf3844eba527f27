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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("unit", ["k_small", "k_syrk_small"])
def test_lds_dma_base_is_never_picked_from_a_lane(tmp_path, unit):
    """global_load_lds takes its LDS base from M0, one value per wave.  A request inside a DIVERGENT branch lets the compiler merge
    differently-masked requests and choose between their (uniform) bases with a v_readfirstlane of a per-lane select: the first active
    lane's base is then used for every lane, and rows land on their neighbours (seen in round 6 with a per-row 'lanes below the row's
    end' staging loop in k_syrk_small - parity broke only for 144 <= mp <= 240 and more than one chunk).  The staging loops keep every
    request under wave-uniform control flow; this guard fails if a compiler or source change brings the pattern back."""
    asm = tmp_path / (unit + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "gpz_amd", "csrc"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "gpz_amd", "csrc", unit + ".hip"), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    lines = [l.strip() for l in asm.read_text().splitlines()]
    assert any(l.startswith("global_load_lds") for l in lines)
    bad = []
    for i, l in enumerate(lines):
        m = re.match(r"s_mov_b32 m0, (s\d+)", l)
        if m and any(re.match(r"v_readfirstlane_b32 %s," % m.group(1), p) for p in lines[max(0, i - 8):i]):
            bad.append((i, lines[max(0, i - 3):i + 2]))
    assert not bad, bad[:3]
