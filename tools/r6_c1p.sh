# round 6: k_conv1_pool after the in-register pre-pooling + 80-float cell stride: parity test, kernel duration (rocprofv3), LDS / MFMA counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6/c1p
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -k "conv1_pool or local_stage_logits" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 10 --warmup 3 --soak-seconds 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/c1p/prof -- $B > gpurun_out/r6/c1p/rocprof.log 2>&1; echo "rocprof rc=$?"
grep -h "k_conv1_pool\|k_conv_pm\|k_maxpool" gpurun_out/r6/c1p/prof/*/*kernel_stats.csv | cut -c1-160
B2="python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 2 --warmup 1 --soak-seconds 0"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r6/c1p/sq1 -- $B2 > gpurun_out/r6/c1p/sq1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/r6/c1p/sq2 -- $B2 > gpurun_out/r6/c1p/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r6/c1p/sq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_conv1_pool" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print({k: round(v) for k, v in m.items()})
if m:
    print("mfma_busy", m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), "lds conflict frac", m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"])
PY
