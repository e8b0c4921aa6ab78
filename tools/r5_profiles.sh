# Round-5 evidence run on the GPU box: bench (full line), rocprofv3 kernel stats of the one-stream headline command, the counter passes
# behind roofline.traffic (tools/pmc_passes.sh -> tools/summarize_pmc.py, which stamps the record with lib/BUILD_INFO.json's
# kernel-source sha and tile shape), the local-training launch census, the end-to-end workflow demo.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5v
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5v/prof -- python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 20 --warmup 5 --soak-seconds 0 > gpurun_out/r5v/rocprof.log 2>&1; echo "rocprof rc=$?"
bash tools/pmc_passes.sh > gpurun_out/r5v/pmc_passes.log 2>&1; echo "pmc rc=$?"
python3 tools/summarize_pmc.py gpurun_out/pmc r05 > gpurun_out/r5v/pmc_summary.log 2>&1; echo "summary rc=$?"; mkdir -p gpurun_out/r5v/profiles && cp profiles/r05_pmc_summary.json profiles/r05_pmc_traffic.json gpurun_out/r5v/profiles/ 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 50 --warmup 5 > gpurun_out/r5v/bench_one_stream.json 2>/dev/null; echo "bench1 rc=$?"
timeout -k 10 700 python bench.py --layers > gpurun_out/r5v/bench.json 2> gpurun_out/r5v/bench_layers.log; echo "bench rc=$?"
bash tools/train_trace.sh r5v_train > gpurun_out/r5v/train_trace.log 2>&1; echo "train trace rc=$?"
DEMO_TRAIN=2000 DEMO_VAL=200 bash tools/workflow_demo.sh > gpurun_out/r5v/workflow_demo.log 2>&1; echo "demo rc=$?"; tail -n 8 gpurun_out/r5v/workflow_demo.log
grep "^{" gpurun_out/r5v/bench.json | cut -c1-300
