# round 6: sweep the balanced launch's cost-model knobs on the graph-replayed step (one box, back to back)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
cd blurry-edges_amd
run() { tag=$1; shift; r=$(env "$@" timeout -k 10 120 python3 -m be_hip.train_local --steps 300 --graph 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"); echo "$tag $* -> $r ms" | tee -a ../gpurun_out/r6/sweep.log; }
: > ../gpurun_out/r6/sweep.log
for ww in 1 1.5 2; do for fc in 0 4; do run snap BE_SK_WW=$ww BE_SK_FC=$fc; run nosnap BE_SK_NO_SNAP=1 BE_SK_WW=$ww BE_SK_FC=$fc; done; done
