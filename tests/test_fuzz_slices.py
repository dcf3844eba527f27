"""Seeded slices of the randomised sweeps under tools/fuzz_*.py, inside the -m gpu suite (the sweeps found real
bugs — validation-only NaN patterns, empty ranks — but ran only by hand).  Each tool is run as the script it is, with a fixed seed
and a case count sized so the five together stay around half a minute; a failure prints the tool's own FAIL / ERROR lines."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tool, arguments (cases, seed[, world]).  Seeds differ from the ones used for the by-hand sweeps recorded in README.md.
SLICES = [
    ("fuzz_parity.py", ["50", "20301"]),
    ("fuzz_predict.py", ["50", "20302"]),
    ("fuzz_f32.py", ["50", "20303"]),
    ("fuzz_mgpu.py", ["50", "20304"]),
    ("fuzz_sharded.py", ["24", "20305", "2"]),
    # d > 20 / k > 8 draws: the runtime-d kernels, the workspace-backed general path, the any-k PHI build (round 3)
    ("fuzz_parity.py", ["40", "20311", "wide"]),
    ("fuzz_predict.py", ["40", "20312", "wide"]),
]


@pytest.mark.parametrize("tool,args", SLICES, ids=[s[0][:-3] + ("_wide" if "wide" in s[1] else "") for s in SLICES])
def test_seeded_fuzz_slice(tool, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, timeout=900,
                       cwd=ROOT)
    out = r.stdout + r.stderr
    m = re.search(r"(\d+) [^\n]*?cases[^\n]*?, (\d+) failures", out)
    assert m, out[-3000:]
    assert int(m.group(1)) == int(args[0]) and int(m.group(2)) == 0 and r.returncode == 0, out[-3000:]
    print(m.group(0), flush=True)
