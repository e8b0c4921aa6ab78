# Round check on the GPU box: all gpu tests, smoke, bench (with cpu baseline), rocprofv3 kernel stats.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/check
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/check/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/check/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/check/smoke.log
timeout -k 10 600 python bench.py --layers > gpurun_out/check/bench.json 2> gpurun_out/check/bench_layers.log; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/check/prof -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/check/rocprof.log 2>&1; echo "rocprof rc=$?"
tail -3 gpurun_out/check/pytest_gpu.log; tail -4 gpurun_out/check/smoke.log; cat gpurun_out/check/bench.json | cut -c1-1500
