# A/B of non-temporal loads / stores in the Winograd transform kernels (BE_WINO_NT = 0..3) and the max-pool (BE_POOL_NT), one stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3nt
for cfg in "0 0" "1 0" "1 1" "0 1" "1 0"; do
  set -- $cfg
  BE_WINO_NT=$1 BE_POOL_NT=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra --streams 1 --steps 40 --warmup 5 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('wino nt $1 pool nt $2: %.1f pairs/s  %.3f ms  dom %.4f ms  hbm %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], [(h['ms_per_iter'], h['achieved_TBps']) for h in d['roofline']['hbm_bound_kernels'][:2]]))
" >> gpurun_out/r3nt/sweep2.log 2>&1 || echo "fail $cfg" >> gpurun_out/r3nt/sweep2.log
done
cat gpurun_out/r3nt/sweep2.log
