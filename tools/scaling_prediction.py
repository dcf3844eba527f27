#!/usr/bin/env python
"""What the first real multi-GPU run should be read against.

Input: single-GPU bench lines of c4 on the rows one of N ranks holds (`bench.py --rows n/N`, measured on this pool's 1-GPU boxes) and the
sizes of the two all-reduces.  Output: profiles/<tag>_scaling_prediction.json - per N the measured shard time, an all-reduce estimate and
the predicted evaluations/s.  The estimate is a MODEL, labelled as such: no collective has run between two devices on this pool.

all-reduce model: ring over xGMI, t = 2 (N-1) alpha + 2 (N-1)/N * bytes / B, alpha = 8 us per hop (RCCL launch + one xGMI hop),
B = 153 GB/s per link and direction (MI355X_MICROARCH.md / SURVEY.md 8e); a direct all-to-all reduce-scatter over the 7 links would be up
to 7 x faster on the bandwidth term - the ring is the conservative bound.
usage: tools/scaling_prediction.py gpurun_out/<dir> profiles/r05_scaling_prediction.json"""
import json
import sys

src, dst = sys.argv[1].rstrip("/"), sys.argv[2]
N_ROWS = 1_000_000
M, D = 1000, 10
MP = 1008
AR1 = 8 * (MP * MP + 4)                       # [PHI'W PHI | PHI'W y | sums]   (comm1: k mp^2 + a few doubles)
AR2 = 8 * (M * (D * D + D + 3) + 2 * MP + 8)  # gradient records + vectors       (comm2)
ALPHA, BW = 8e-6, 153e9


def allreduce(nbytes, n):
    return 0.0 if n == 1 else 2 * (n - 1) * ALPHA + 2.0 * (n - 1) / n * nbytes / BW


def line(path):
    with open(path) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


out = {"workload": "c4: n=1e6 d=10 m=1000 VC heteroscedastic fp64, rows split over N ranks, two all-reduces per evaluation",
       "status": "PREDICTION from single-GPU shard measurements + a ring all-reduce model; no N > 1 run exists on this pool",
       "allreduce_bytes": {"exchange_1": AR1, "exchange_2": AR2},
       "allreduce_model": "ring: 2 (N-1) * 8 us + 2 (N-1)/N * bytes / 153 GB/s",
       "per_n": []}
for n, name in ((1, "c4.json"), (2, "c4_shard500k.json"), (4, "c4_shard250k.json"), (8, "c4_shard125k.json")):
    try:
        d = line(f"{src}/{name}")
    except Exception as e:  # noqa: BLE001
        out["per_n"].append({"n_gpus": n, "error": repr(e)})
        continue
    shard_ms = d["ms_per_step"]
    ar_ms = (allreduce(AR1, n) + allreduce(AR2, n)) * 1e3
    stages = d.get("kernels", {}).get("stage_ms_per_eval", {})
    chain = sum(stages.get(k, 0.0) for k in ("chol", "trtri", "lauum", "solve_vectors", "syrk_reduce", "finish", "unpack", "row_sums"))
    out["per_n"].append({"n_gpus": n, "rows_per_gpu": N_ROWS // n, "measured_shard_ms_single_gpu": shard_ms,
                         "replicated_m_by_m_chain_ms": chain, "allreduce_estimate_ms": ar_ms,
                         "predicted_ms_per_eval": shard_ms + ar_ms, "predicted_evals_per_s": 1e3 / (shard_ms + ar_ms),
                         "predicted_efficiency_vs_1gpu": None,
                         "stage_ms_per_eval": stages, "route": d.get("route")})
base = next((p for p in out["per_n"] if p.get("n_gpus") == 1 and "predicted_ms_per_eval" in p), None)
if base:
    for p in out["per_n"]:
        if "predicted_ms_per_eval" in p:
            p["predicted_efficiency_vs_1gpu"] = base["predicted_ms_per_eval"] / (p["n_gpus"] * p["predicted_ms_per_eval"])
# config 5 (n = 2e6, d = 20, m = 2000, VC + diagonal Psi, dtype f32): the full problem on one GPU and the rows one of 8 ranks holds
M5, D5, MP5 = 2000, 20, 2016
AR1_5 = 8 * (MP5 * MP5 + 4)
AR2_5 = 8 * (M5 * (D5 * D5 + D5 + 3) + 2 * MP5 + 8)
c5 = {"workload": "c5: n=2e6 d=20 m=2000 VC heteroscedastic + diagonal Psi, dtype f32 (fp32 pair kernels, fp32-operand MFMA contractions)",
      "status": "PREDICTION from single-GPU measurements + the ring all-reduce model above; no N > 1 run exists on this pool",
      "allreduce_bytes": {"exchange_1": AR1_5, "exchange_2": AR2_5}, "per_n": []}
for n, name in ((1, "c5_full_1gpu.json"), (8, "c5s.json")):
    try:
        d = line(f"{src}/{name}")
    except Exception as e:  # noqa: BLE001
        c5["per_n"].append({"n_gpus": n, "error": repr(e)})
        continue
    ar_ms = (allreduce(AR1_5, n) + allreduce(AR2_5, n)) * 1e3
    c5["per_n"].append({"n_gpus": n, "rows_per_gpu": 2_000_000 // n, "measured_ms_single_gpu": d["ms_per_step"], "allreduce_estimate_ms": ar_ms,
                        "predicted_ms_per_eval": d["ms_per_step"] + ar_ms, "predicted_evals_per_s": 1e3 / (d["ms_per_step"] + ar_ms),
                        "stage_ms_per_eval": d.get("kernels", {}).get("stage_ms_per_eval", {})})
b5 = next((q for q in c5["per_n"] if q.get("n_gpus") == 1 and "predicted_ms_per_eval" in q), None)
for q in c5["per_n"]:
    if b5 and "predicted_ms_per_eval" in q:
        q["predicted_efficiency_vs_1gpu"] = b5["predicted_ms_per_eval"] / (q["n_gpus"] * q["predicted_ms_per_eval"])
out["c5"] = c5
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({"per_n": [{k: p.get(k) for k in ("n_gpus", "measured_shard_ms_single_gpu", "allreduce_estimate_ms", "predicted_evals_per_s",
                                                   "predicted_efficiency_vs_1gpu")} for p in out["per_n"]],
                  "c5": [{k: q.get(k) for k in ("n_gpus", "measured_ms_single_gpu", "allreduce_estimate_ms", "predicted_evals_per_s",
                                                "predicted_efficiency_vs_1gpu")} for q in c5["per_n"]]}, indent=1))
