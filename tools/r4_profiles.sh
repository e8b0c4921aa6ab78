# Round-4 evidence run on the GPU box: bench (full), rocprofv3 kernel stats of the one-stream headline command, local / global training traces.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4v
timeout -k 10 700 python bench.py --layers > gpurun_out/r4v/bench.json 2> gpurun_out/r4v/bench_layers.log; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 50 --warmup 5 > gpurun_out/r4v/bench_one_stream.json 2>/dev/null; echo "bench1 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4v/prof -- python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 20 --warmup 5 > gpurun_out/r4v/rocprof.log 2>&1; echo "rocprof rc=$?"
bash tools/train_trace.sh r4v_train
export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
cd blurry-edges_amd && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/r4v/global -- python3 -m be_hip.train_global --steps 20 --images 8 --batch 8 > ../gpurun_out/r4v/global.log 2>&1; echo "global rc=$?"; cd ..
python3 tools/global_trace.py gpurun_out/r4v/global 40 > gpurun_out/r4v/global_step.txt 2>&1
grep "^{" gpurun_out/r4v/bench.json | cut -c1-400
