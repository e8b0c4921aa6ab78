cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ptrain
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
cd blurry-edges_amd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/ptrain/local -- python3 -m be_hip.train_local --steps 100 > ../gpurun_out/ptrain/local.log 2>&1; echo "local rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/ptrain/global -- python3 -m be_hip.train_global --steps 20 --images 8 --batch 8 > ../gpurun_out/ptrain/global.log 2>&1; echo "global rc=$?"
tail -1 ../gpurun_out/ptrain/local.log; tail -1 ../gpurun_out/ptrain/global.log
