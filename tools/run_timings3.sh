cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tl3
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_dp_gloo.py -q > gpurun_out/tl3/pytest_train.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/tl3/pytest_train.log)"
timeout -k 10 200 python -m be_hip.train_local --steps 300 --graph > gpurun_out/tl3/train_local_graph.log 2>&1; echo "tl rc=$?"
timeout -k 10 200 python -m be_hip.train_local --steps 300 > gpurun_out/tl3/train_local_eager.log 2>&1; echo "tl rc=$?"
timeout -k 10 200 python -m be_hip.train_global --steps 40 --images 8 --batch 8 > gpurun_out/tl3/train_global_b8.log 2>&1; echo "tg rc=$?"
tail -1 gpurun_out/tl3/train_local_graph.log; tail -1 gpurun_out/tl3/train_local_eager.log; tail -1 gpurun_out/tl3/train_global_b8.log
