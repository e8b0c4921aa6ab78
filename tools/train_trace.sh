# rocprofv3 kernel trace of the local training step (eager launches): per-kernel stats + the launch sequence of one step.
# usage (on the GPU box): bash tools/train_trace.sh <tag>
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
cd blurry-edges_amd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/$TAG/local -- python3 -m be_hip.train_local --steps 40 > ../gpurun_out/$TAG/local.log 2>&1; echo "local rc=$?"
tail -1 ../gpurun_out/$TAG/local.log
cd .. && python3 tools/train_trace.py gpurun_out/$TAG/local > gpurun_out/$TAG/step_sequence.txt 2>&1; tail -5 gpurun_out/$TAG/step_sequence.txt
timeout -k 10 200 python3 -m be_hip.train_local --steps 200 --graph > gpurun_out/$TAG/graph.log 2>&1; tail -1 gpurun_out/$TAG/graph.log
