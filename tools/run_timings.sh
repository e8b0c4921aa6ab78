cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tl2
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 300 python tools/time_pipeline.py > gpurun_out/tl2/pipeline.log 2>&1; echo "pipeline rc=$?"
timeout -k 10 200 python -m be_hip.train_global --steps 10 --images 8 --batch 8 > gpurun_out/tl2/train_global_b8.log 2>&1; echo "tg rc=$?"
timeout -k 10 200 python -m be_hip.train_local --steps 200 --graph > gpurun_out/tl2/train_local_graph.log 2>&1; echo "tl rc=$?"
tail -4 gpurun_out/tl2/pipeline.log; tail -3 gpurun_out/tl2/train_global_b8.log; tail -3 gpurun_out/tl2/train_local_graph.log
