# Round-6 evidence run on the GPU box (every figure DESIGN section 3 quotes comes from a file this writes; copy gpurun_out/r6v/* into
# profiles/ as r06_*): the full bench line, rocprofv3 kernel stats of the one-stream headline command, the counter passes behind
# roofline.traffic and the per-kernel MFMA-busy / LDS figures (inference AND the local training step), the local-training launch
# census, kernel stats of the global-stage training step.   usage: bash tools/r6_profiles.sh [tag]
TAG=${1:-r6v}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 20 --warmup 5 --soak-seconds 0 > gpurun_out/$TAG/rocprof.log 2>&1; echo "rocprof rc=$?"
cp gpurun_out/$TAG/prof/*/*kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/pmc gpurun_out/pmc_train
bash tools/pmc_passes.sh > gpurun_out/$TAG/pmc_passes.log 2>&1; echo "pmc rc=$?"
python3 tools/summarize_pmc.py gpurun_out/pmc r06 > gpurun_out/$TAG/pmc_summary.log 2>&1; echo "summary rc=$?"
cp profiles/r06_pmc_summary.json profiles/r06_pmc_traffic.json gpurun_out/$TAG/ 2>/dev/null
bash tools/pmc_train.sh > gpurun_out/$TAG/pmc_train.log 2>&1; echo "pmc train rc=$?"
python3 tools/summarize_pmc.py gpurun_out/pmc_train r06_train > gpurun_out/$TAG/pmc_train_summary.log 2>&1; echo "train summary rc=$?"
cp profiles/r06_train_pmc_summary.json gpurun_out/$TAG/ 2>/dev/null
rm -rf gpurun_out/pmc gpurun_out/pmc_train         # the raw passes are large; the summaries are what is kept
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 50 --warmup 5 > gpurun_out/$TAG/bench_one_stream.json 2>/dev/null; echo "bench1 rc=$?"
timeout -k 10 900 python bench.py --layers > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench_layers.log; echo "bench rc=$?"
bash tools/train_trace.sh ${TAG}_train > gpurun_out/$TAG/train_trace.log 2>&1; echo "train trace rc=$?"
cp gpurun_out/${TAG}_train/step_sequence.txt gpurun_out/$TAG/local_train_step_sequence.txt 2>/dev/null
cp gpurun_out/${TAG}_train/local/*/*kernel_stats.csv gpurun_out/$TAG/local_train_kernel_stats.csv 2>/dev/null
tail -1 gpurun_out/${TAG}_train/graph.log > gpurun_out/$TAG/local_train_graph.json 2>/dev/null
bash tools/prof_global_train.sh > gpurun_out/$TAG/global_train.log 2>&1; echo "global rc=$?"
cp gpurun_out/pglobal/prof/*/*kernel_stats.csv gpurun_out/$TAG/global_train_b8_kernel_stats.csv 2>/dev/null
cp gpurun_out/pglobal/plain.log gpurun_out/$TAG/global_train_b8_plain.log 2>/dev/null
rm -rf gpurun_out/$TAG/prof gpurun_out/${TAG}_train/local gpurun_out/pglobal/prof
grep "^{" gpurun_out/$TAG/bench.json | cut -c1-300
