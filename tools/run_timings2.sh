cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tl2
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 200 python -m be_hip.train_global --steps 40 --images 8 --batch 8 > gpurun_out/tl2/tg_new.log 2>&1; echo "tg rc=$?"
BE_NO_GEMM_ROWS=1 timeout -k 10 200 python -m be_hip.train_global --steps 40 --images 8 --batch 8 > gpurun_out/tl2/tg_old.log 2>&1; echo "tg rc=$?"
timeout -k 10 200 python -m be_hip.train_global --steps 40 --images 8 --batch 8 > gpurun_out/tl2/tg_new2.log 2>&1; echo "tg rc=$?"
tail -1 gpurun_out/tl2/tg_new.log; tail -1 gpurun_out/tl2/tg_old.log; tail -1 gpurun_out/tl2/tg_new2.log
