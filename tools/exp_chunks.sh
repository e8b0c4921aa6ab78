cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp1
for c in 8192 4096 2048; do timeout -k 10 200 python bench.py --no-cpu-baseline --chunk $c --layers > gpurun_out/exp1/chunk$c.log 2>&1 || exit 1; done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/exp1/sq1 -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/exp1/sq1.log 2>&1
grep -h '"value"' gpurun_out/exp1/chunk*.log | cut -c1-160
