#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by tools/pmc_passes.sh into profiles/<tag>_pmc_summary.json and
profiles/<tag>_pmc_traffic.json (usage: summarize_pmc.py <dir> <tag> [git head]; HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md's HBM
section prescribes: FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on gfx950 -> x2;
WRITE_SIZE in KiB is exact for 16-B streaming stores; our epilogue stores are 4-B per lane, 128-B segments, so the
write side is 'uncalibrated width' and reported as is)."""
import collections
import csv
import glob
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
NAMES = {"k_conv_igemm<2, 2, 2, 2, 0,": "conv128x128", "k_conv_igemm<4, 1, 1, 3, 0,": "conv128x96",
         "k_conv_igemm<4, 1, 1, 2, 1,": "conv1_row8", "k_render_colors": "render_pass_a",
         "k_wino_in": "wino_in", "k_wino_out_in": "wino_out_in", "k_wino_out(": "wino_out", "k_wino_out_pool2": "wino_out_pool2",
         "k_wino_gemm_ws<6, 1>": "gemm_rows_ws", "k_wino_gemm_ws<16, 1>": "gemm_rows_ws", "k_wino_gemm_ws<24, 1>": "gemm_rows_ws",
         "k_wino_gemm_ws": "wino_gemm", "k_wino_gemm<0>": "wino_gemm", "k_wino_gemm<1>": "gemm_rows", "k_wino_gemm(": "wino_gemm",
         "k_conv1_pool": "conv1_pool_image_major", "k_conv_pm<2, 3, 0>": "conv_pm_256x96", "k_conv_pm<2, 2, 1>": "conv_pm_conv1", "k_maxpool_nhwc": "maxpool",
         # the training step (tools/pmc_train.sh)
         "k_unit_gemms_sk": "train_unit_gemms_balanced", "k_unit_gemms<0>": "train_unit_gemms_64x64", "k_unit_gemms<1>": "train_unit_gemms_128x32", "k_unit_gemms<2>": "train_unit_gemms_64x64_uniform", "k_conv_igemm<2, 2, 1, 1, 0,": "train_conv_64x64",
         "k_conv_igemm<4, 1, 1, 1, 0,": "train_conv_128x32", "k_bwd_post": "train_bwd_post", "k_bn_stats": "train_bn_stats",
         "k_bn_fwd_apply": "train_bn_fwd_apply", "k_bn_bwd_reduce": "train_bn_bwd_reduce", "k_bn_bwd_apply": "train_bn_bwd_apply",
         "k_local_loss(": "train_local_loss", "k_clip_adamw": "train_clip_adamw", "k_pack_jobs": "train_pack_jobs"}


def short(name):
    for k, v in NAMES.items():
        if k in name:
            return v
    return None


summary = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{src}/*/")):
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                summary[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in summary.items()}
for k, cs in out.items():
    if "GRBM_GUI_ACTIVE" in cs and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
        cyc = cs["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
        cs["derived"] = {"mfma_busy_frac": cs["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024 / cyc,
                         "wait_any_frac": cs["SQ_WAIT_ANY"]["mean_per_launch"] / cs["SQ_WAVE_CYCLES"]["mean_per_launch"],
                         "wait_inst_frac": cs["SQ_WAIT_INST_ANY"]["mean_per_launch"] / cs["SQ_WAVE_CYCLES"]["mean_per_launch"]}
    if "TCC_HIT_sum" in cs:
        h, m = cs["TCC_HIT_sum"]["mean_per_launch"], cs["TCC_MISS_sum"]["mean_per_launch"]
        cs.setdefault("derived", {})["l2_hit_rate"] = h / (h + m)
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
head = sys.argv[3] if len(sys.argv) > 3 else None
# the sources and tile shape the measured library was built from: recorded by build() in lib/BUILD_INFO.json (travels to the GPU box
# with the libraries); bench.py fills roofline.traffic from this record only while the RUNNING library's BUILD_INFO names the same
import os
_info_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blurry-edges_amd", "lib", "BUILD_INFO.json")
INFO = json.load(open(_info_path)) if os.path.exists(_info_path) else {}
head = head or INFO.get("git_head")


dom = out.get("wino_gemm") or out.get("conv128x128", {})
if "FETCH_SIZE" in dom and "WRITE_SIZE" in dom:
    rd = dom["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    wr = dom["WRITE_SIZE"]["mean_per_launch"] * 1024
    wino = "wino_gemm" in out
    # bench.py reports this under roofline.traffic_recorded only while kernel_id still names its dominant kernel
    json.dump({"kernel": "k_wino_gemm_ws<6|16|24> (be_wino.hip)" if wino else "k_conv_igemm<2,2,2,2,TAPS>", "kernel_id": 6 if wino else 0,
               "bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
               # positions x 4 B x (tiles x cin + cin x cout + tiles x cout) averaged over the six launches of a step, n = 8192:
               # 40 positions x 2 n tiles (8x5 Winograd tiles, round 4); 25 x 4 n gave 1878289066.67
               "algo_bytes_per_launch": 1509294080.0 if wino else None, "git_head": head,
               # bench.py fills roofline.traffic from this record only while these sources are byte-identical to the running library's
               "kernel_sources": INFO.get("kernel_sources"), "kernel_source_sha": INFO.get("kernel_source_sha"),
               "wino_tile_rows": INFO.get("wino_tile_rows"),
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_passes.sh); FETCH_SIZE x2 (gfx950), KiB -> B; "
                         "counts L2 misses, i.e. Infinity-Cache hits too",
               "launches_averaged": dom["FETCH_SIZE"]["launches"]}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in out.items()}, indent=1))
