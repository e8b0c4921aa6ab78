# PMC passes over the GlobalLoss kernel at batch 8 (tools/bench_global_loss.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_gl
python3 tools/bench_global_loss.py 8 10 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_gl/a -- python3 tools/bench_global_loss.py 8 3 > gpurun_out/pmc_gl/a.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_gl/b -- python3 tools/bench_global_loss.py 8 3 > gpurun_out/pmc_gl/b.log 2>&1
ls gpurun_out/pmc_gl/*/*/ | head -4
