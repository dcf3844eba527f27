"""Developer tool: k_syrk time at c4 over (GPZ_SYRK_S1, GPZ_SYRK_S2) = row ranges of the off-diagonal / diagonal tiles."""
import json, os, subprocess, sys
pairs = [tuple(map(int, a.split(","))) for a in sys.argv[1:]] or [(30, 23), (31, 19), (31, 21), (32, 16), (32, 18), (32, 20), (33, 14), (34, 16)]
for s1, s2 in pairs:
    env = dict(os.environ, GPZ_SYRK_S1=str(s1), GPZ_SYRK_S2=str(s2))
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "10"], capture_output=True, text=True, env=env).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print("S1=%d S2=%d  WGs=%d  syrk %.3f ms  step %.3f ms" % (s1, s2, 28 * s1 + 8 * s2, d["kernels"]["stage_ms_per_eval"]["syrk"], d["ms_per_step"]), flush=True)
