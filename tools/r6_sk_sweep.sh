# round 6: sweep the balanced launch's cost-model knobs on the graph-replayed step (one box, back to back)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_train_gpu.py -x -q -k "training_unit_matches or unit_pair_launches" 2>&1 | tail -3
cd blurry-edges_amd
run() { tag=$1; shift; r=$(env "$@" timeout -k 10 120 python3 -m be_hip.train_local --steps 300 --graph 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"); echo "$tag $* -> $r ms" | tee -a ../gpurun_out/r6/sweep.log; }
: > ../gpurun_out/r6/sweep.log
for ww in 0.8 1 1.25 1.5 2 2.5; do run ww BE_SK_WW=$ww; done
for ww in 1 1.5 2; do run nofwd BE_NO_TRAIN_SK_FWD=1 BE_SK_WW=$ww; done
run old BE_NO_TRAIN_SK=1
