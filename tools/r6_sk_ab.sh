# round 6: balanced backward launch (k_unit_gemms_sk) - unit tests, then the graph-replayed step with and without it
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -k "training_unit_matches or unit_pair_launches" > gpurun_out/r6/sk_unit_tests.log 2>&1; echo "unit tests rc=$?"; tail -5 gpurun_out/r6/sk_unit_tests.log
cd blurry-edges_amd
timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph > ../gpurun_out/r6/graph_sk.log 2>&1; tail -1 ../gpurun_out/r6/graph_sk.log
BE_NO_TRAIN_SK=1 timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph > ../gpurun_out/r6/graph_old.log 2>&1; tail -1 ../gpurun_out/r6/graph_old.log
timeout -k 10 200 python3 -m be_hip.train_local --steps 400 --graph > ../gpurun_out/r6/graph_sk2.log 2>&1; tail -1 ../gpurun_out/r6/graph_sk2.log
