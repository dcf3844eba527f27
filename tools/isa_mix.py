#!/usr/bin/env python3
"""Developer tool: static instruction mix per kernel of a hipcc -S listing (mnemonic classes, scratch traffic, registers).
   usage: tools/isa_mix.py build/k_psi32.s [substring-of-kernel-name]"""
import re, sys, collections
path = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None; mix = {}
meta = collections.defaultdict(dict)
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur = m.group(1); mix[cur] = collections.Counter(); continue
    m = re.match(r"\s*\.set (_Z\w+)\.(num_vgpr|num_agpr|private_seg_size|numbered_sgpr), (\d+)", line)
    if m: meta[m.group(1)][m.group(2)] = int(m.group(3)); continue
    if cur is None: continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"): cur = None; continue
    m = re.match(r"\s+([a-z_0-9]+)\b", line)
    if not m or line.strip().startswith((".", ";")): continue
    op = m.group(1)
    if op.startswith("v_pk_"): c = "v_pk"
    elif op.startswith("v_mfma"): c = "mfma"
    elif op.startswith("v_accvgpr"): c = "v_accvgpr"
    elif re.search(r"_dpp|permlane|readlane|writelane|v_swap", line): c = "v_xlane"
    elif op.startswith("v_mov") : c = "v_mov"
    elif op.startswith(("v_cndmask", "v_bfi", "v_and", "v_or", "v_lshl", "v_lshr", "v_cmp")): c = "v_bit/sel"
    elif op.startswith(("v_rsq", "v_rcp", "v_log", "v_exp", "v_sqrt")): c = "v_trans"
    elif re.match(r"v_.*_f64", op): c = "v_f64"
    elif op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac", "v_mac")): c = "v_f32"
    elif op.startswith("v_"): c = "v_other"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith("scratch_"): c = "scratch"
    elif op.startswith(("global_", "flat_", "buffer_")): c = "vmem"
    elif op.startswith("s_nop"): c = "s_nop"
    elif op.startswith("s_waitcnt"): c = "s_waitcnt"
    elif op.startswith("s_"): c = "salu"
    else: c = "other"
    mix[cur][c] += 1
for k, c in mix.items():
    if filt not in k or not c: continue
    valu = sum(v for kk, v in c.items() if kk.startswith("v_"))
    print(f"{k}\n  regs {meta.get(k)}  total {sum(c.values())}  VALU {valu}")
    print("  " + "  ".join(f"{kk}:{v}" for kk, v in sorted(c.items(), key=lambda t: -t[1])))
