# global-stage training step (batch 8, 147 x 147 pairs): un-profiled timing first, then rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pglobal
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
cd blurry-edges_amd
timeout -k 10 300 python3 -m be_hip.train_global --steps 40 --images 8 --batch 8 > ../gpurun_out/pglobal/plain.log 2>&1 && tail -1 ../gpurun_out/pglobal/plain.log &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/pglobal/prof -- python3 -m be_hip.train_global --steps 20 --images 8 --batch 8 > ../gpurun_out/pglobal/prof.log 2>&1; echo "prof rc=$?"
tail -1 ../gpurun_out/pglobal/prof.log
