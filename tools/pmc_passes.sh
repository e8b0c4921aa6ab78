cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
B="python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 2 --warmup 1 --soak-seconds 0"   # one stream: a kernel's counters and duration mean something only when it has the chip to itself
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc/$name -- $B > gpurun_out/pmc/$name.log 2>&1 || echo "FAILED $name"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
ls gpurun_out/pmc/*/* | head; tail -3 gpurun_out/pmc/*.log | cut -c1-300
