cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_hip_parity.py -x -q -k "train or grad or unit or teacher or free_running or graph" > gpurun_out/r6/sk_tests4.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6/sk_tests4.log
cd blurry-edges_amd
for i in 1 2; do timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph 2>/dev/null | tail -1 | cut -c1-140; done
BE_NO_TRAIN_SK=1 timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph 2>/dev/null | tail -1 | cut -c1-140
