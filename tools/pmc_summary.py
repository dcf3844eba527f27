"""Summarise the counter passes written by tools/pmc_run.sh into a text file under profiles/, and (optionally) write the
per-launch fabric-side byte counts bench.py reports as `roofline.traffic` into profiles/pmc_constants.json.
usage: python tools/pmc_summary.py gpurun_out/<dir> profiles/<name>.txt ["bench arguments and workload, for the header"]
                                   [--constants profiles/pmc_constants.json --config c4]"""
import collections
import csv
import json
import os
import sys

argv = list(sys.argv[1:])
constants = config = None
if "--constants" in argv:
    i = argv.index("--constants"); constants = argv[i + 1]; del argv[i:i + 2]
if "--config" in argv:
    i = argv.index("--config"); config = argv[i + 1]; del argv[i:i + 2]
base, dst = argv[0].rstrip("/") + "/", argv[1]
what = argv[2] if len(argv) > 2 else ("--steps 2 --warmup 2 --no-cpu-baseline\n"
                                      "# (c4: n=1e6 d=10 m=1000 VC hetero, 1 x MI355X")


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


data = {}
for f in ("sq", "sq2", "fetch", "write"):
    for k, cs in load(base + f + "/p_counter_collection.csv").items():
        for c, v in cs.items():
            data.setdefault(k, {})[c] = sum(v) / len(v)
# average duration per kernel from the kernel-trace pass of the same command (trace/t_kernel_stats.csv), when it is there
dur_ns = {}
try:
    for r in csv.DictReader(open(base + "trace/t_kernel_stats.csv")):
        dur_ns[r["Name"].split("(")[0]] = float(r["AverageNs"])
except Exception:
    pass
keys = [k for k in data if k.startswith(("k_tgemm", "void k_tgemm", "void k_syrk<true", "void k_phi", "void k_moments", "void k_row_epilogue", "void k_psi32", "void k_psi_", "void k_small_tail", "void k_syrk_small<"))]
with open(dst, "w") as out:
    out.write("# rocprofv3 --pmc <group> --kernel-trace -- python bench.py " + what + "; one counter group per pass; tools/pmc_run.sh)\n"
              "# per-launch averages.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles;\n"
              "# SQ_VALU_MFMA_BUSY_CYCLES in cycles (= 64 x N_mfma for v_mfma_f64_16x16x4_f64); GRBM_GUI_ACTIVE is summed over\n"
              "# the 8 XCDs (divide by 8 for kernel cycles); FETCH_SIZE / WRITE_SIZE in KB at the L2 fabric side; FETCH_SIZE reads\n"
              "# 1/2 of the bytes of wide streaming loads on gfx950 (MI355X_MICROARCH.md, HBM section) -> 'x2 corrected'.\n")
    for k in sorted(keys):
        d = data[k]
        out.write("\n%s\n" % k)
        for c in sorted(d):
            out.write("   %-30s %.6g\n" % (c, d[c]))
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
        if d.get("SQ_INSTS_MFMA", 0) > 0 and cyc > 0:
            util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
            out.write("   -> kernel cycles (GRBM/8) %.4g; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) = %.1f %%\n"
                      % (cyc, 100 * util))
            if d.get("SQ_INSTS_VALU", 0) > 0:   # SQ_INSTS_VALU counts the MFMAs too (profiles/r05_pmc_counter_semantics.txt)
                out.write("   -> vector instructions that are not MFMAs, per MFMA: (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA = %.2f\n"
                          % ((d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"]))
        if "FETCH_SIZE" in d:
            out.write("   -> fabric-side bytes per launch: fetch %.2f GB (x2 corrected %.2f GB), write %.2f GB\n"
                      % (d["FETCH_SIZE"] * 1024 / 1e9, 2 * d["FETCH_SIZE"] * 1024 / 1e9, d.get("WRITE_SIZE", 0) * 1024 / 1e9))
            if k in dur_ns and dur_ns[k] > 0:
                tot = (2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024
                out.write("   -> average duration %.1f us (kernel trace of the same command): %.2f TB/s at the fabric (x2-corrected fetch + write)\n"
                          % (dur_ns[k] / 1e3, tot / dur_ns[k] / 1e3))
                if cyc > 0:
                    out.write("   -> kernel clock = kernel cycles / duration = %.2f GHz\n" % (cyc / dur_ns[k]))
print(open(dst).read())

if constants and config:
    # bytes per launch at the L2 fabric side = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, in KB
    def fabric(prefixes):
        for k in sorted(keys):
            if k.startswith(prefixes) and "FETCH_SIZE" in data[k]:
                return (2.0 * data[k]["FETCH_SIZE"] + data[k].get("WRITE_SIZE", 0.0)) * 1024.0, k
        return None, None
    block = {"source": dst + " (rocprofv3 PMC passes of the same command, tools/pmc_run.sh; written by tools/pmc_summary.py)",
             "note": "L2 fabric-side bytes per launch (include Infinity-Cache hits): 2 x FETCH_SIZE (gfx950 correction of "
                     "MI355X_MICROARCH.md, HBM section) + WRITE_SIZE; bench arguments: " + what.splitlines()[0]}
    for name, pre in (("tgemm", ("k_tgemm", "void k_tgemm", "void k_small_tail")), ("syrk", ("void k_syrk<true", "void k_syrk_small<")), ("phi", ("void k_phi", "void k_psi32_phi")),
                      ("moments", ("void k_moments", "void k_psi32_moments"))):
        b, k = fabric(pre)
        if b is not None:
            block[name + "_bytes_per_launch"] = b
            block[name + "_kernel"] = k
            d = data[k]
            cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
            if d.get("SQ_INSTS_MFMA", 0) > 0 and cyc > 0:   # counter-based MFMA utilisation (busy cycles / SIMDs / kernel cycles)
                block[name + "_mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
                block[name + "_valu_per_mfma"] = (d.get("SQ_INSTS_VALU", 0) - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"]
                block[name + "_mfma_instructions"] = d["SQ_INSTS_MFMA"]
            if cyc > 0 and k in dur_ns and dur_ns[k] > 0:          # the clock the kernel actually ran at (GRBM cycles / kernel-trace duration)
                block[name + "_kernel_clock_ghz"] = cyc / dur_ns[k]
    allc = {}
    if os.path.exists(constants):
        allc = json.load(open(constants))
    allc[config] = block
    json.dump(allc, open(constants, "w"), indent=1)
    print("wrote", constants, config, {k: v for k, v in block.items() if k.endswith("per_launch")})

