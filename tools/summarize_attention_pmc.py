#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_attention.sh (lab/bin/attn_lab: the attention kernels of GlobalStage at batch 8)
into <out json>: per kernel MFMA-busy, wait fractions, VALU instructions per MFMA.  usage: summarize_attention_pmc.py [dir] [out]"""
import collections
import csv
import glob
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_attn"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06_attention_pmc_summary.json"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = ("attention_fwd_train" if "k_attention<true" in n else "attention_fwd_eval" if "k_attention<false" in n
             else "attention_bwd_fused" if "k_attn_bwd_fused" in n else None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"launches": len(next(iter(cs.values())))}
    if "GRBM_GUI_ACTIVE" in m:
        d["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        d["wait_inst_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        d["wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in m and m.get("SQ_INSTS_VALU_MFMA_MOPS_F32"):
        mf = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 4.0          # v_mfma_f32_16x16x4_f32 = 2048 FLOP = 4 MOPS of 512
        d["valu_instructions_per_mfma"] = (m["SQ_INSTS_VALU"] - mf) / mf
    out[k] = {"derived": d, "mean_per_launch": m}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out.items()}, indent=1))
