# A/B of one environment knob on the bench: tools/ab_env.sh NAME VALUE_A VALUE_B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for v in $2 $3 $2 $3; do
  env $1=$v timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab/$1_$v.json 2> gpurun_out/ab/$1_$v.err || exit 1
  python -c "import json;j=json.load(open('gpurun_out/ab/$1_$v.json'));print('$1=$v', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
done
