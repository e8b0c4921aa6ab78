cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export PYTHONPATH=$GRAFT_REPO_ROOT/blurry-edges_amd:$GRAFT_REPO_ROOT
for v in 1 0 1 0; do BE_FUSED_ADAMW=$v python -m be_hip.train_local --steps 300 --graph 2>/dev/null | python -c "import json,sys;j=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('fused=$v', j['ms_per_step'], j['last_loss'])"; done
