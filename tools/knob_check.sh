# the A/B knobs still give green parity tests (each selects an older kernel family)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/knobs
for kv in BE_WINOGRAD=0 BE_NO_CONV_PM=1 BE_NO_GEMM_ROWS=1 BE_WINO_NO_PERSIST=1 BE_WINO_NO_WS=1; do
  env $kv timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "logits or ragged or conv_layers" > gpurun_out/knobs/$kv.log 2>&1
  echo "$kv rc=$? $(tail -1 gpurun_out/knobs/$kv.log)"
done
