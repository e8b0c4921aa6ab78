# round 6: balanced persistent launch (k_unit_gemms_sk) - the tests that see it, then the graph-replayed step with and without it
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -k "training_unit_matches or unit_pair_launches or teacher_forced or free_running or graph_replayed or segmented" > gpurun_out/r6/sk_unit_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r6/sk_unit_tests.log
cd blurry-edges_amd
for i in 1 2; do
timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph 2>/dev/null | tail -1 | cut -c1-140
done
BE_NO_TRAIN_SK=1 timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph 2>/dev/null | tail -1 | cut -c1-140
cd .. && bash tools/train_trace.sh r6/trace_sk3 > gpurun_out/r6/trace_sk3.log 2>&1; grep "k_unit_gemms\|k_conv_igemm\|launches" gpurun_out/r6/trace_sk3/step_sequence.txt | cut -c1-100
