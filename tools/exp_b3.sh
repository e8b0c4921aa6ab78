# A/B of the opt-in split-bf16 convolution mode against the default fp32 MFMA
mkdir -p gpurun_out/b3
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "split_bf16" -s > gpurun_out/b3/pytest.log 2>&1 || { tail -20 gpurun_out/b3/pytest.log; exit 1; }
grep "logits vs" gpurun_out/b3/pytest.log
for p in f32 bf16x3; do
  BE_CONV_PRECISION=$p timeout -k 10 200 python bench.py --steps 20 --warmup 5 --layers > gpurun_out/b3/b_$p.json 2> gpurun_out/b3/l_$p.log || exit 1
  python -c "import json; d=json.load(open('gpurun_out/b3/b_$p.json')); print('$p', d['value'], d['ms_per_step'], 'rmse', d.get('depth_rmse_vs_oracle_m'), 'logits', d.get('logits_relmax_vs_oracle'))"
  sed -n 3,13p gpurun_out/b3/l_$p.log
done
