cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/expv
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "conv_layers or logits" > gpurun_out/expv/parity0.log 2>&1 || exit 1
for v in 32 0; do
  BE_CONV_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline --layers > gpurun_out/expv/v$v.log 2>&1 || exit 1
done
BE_CONV_VARIANT=0 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "conv_layers or logits" > gpurun_out/expv/parity1.log 2>&1
tail -2 gpurun_out/expv/parity*.log
for v in 32 0; do echo "== variant $v"; python - <<PY
import json
l=open('gpurun_out/expv/v$v.log').read().strip().split('\n')
j=json.loads(l[-1]); print(j['value'], j['roofline']['achieved'], j['roofline']['frac'])
PY
done
