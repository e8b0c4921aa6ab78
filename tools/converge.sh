# Train to convergence on one MI355X and evaluate (VERDICT r4 "next" #2).  Runs on the GPU box in stages that each fit one gpurun call
# (<= 20 min); what a later stage needs from an earlier one travels as files: a stage writes gpurun_out/r05_converged/, the
# builder copies the checkpoints into ckpt/ (git-ignored, NOT gpurun-ignored) before the next call.  The data set is regenerated
# per call (deterministic, ~20 s for the reference's 8000 + 2000 images).
#   STAGE=local   datagen (utils/args.py:18-24 sizes) -> local_train at the reference's full schedule (args.py:29-36: 1000 epochs)
#   STAGE=global  datagen -> global_pre with ckpt/pretrained_local_stage.pth -> global_train (GLOBAL_EPOCHS, GLOBAL_DYN; resumes
#                 from ckpt/global_resume.ckpt when present)
#   STAGE=eval    held-out synthetic set (seed shifted) -> HIP pipeline metrics, oracle pipeline metrics on the same checkpoint + images
set -e
# ckpt/frozen/ (git-ignored, travels with the snapshot): a copy of the tree + built libraries at the commit the run STARTED on, so
# that all stages of one training run execute the same code while the working tree moves on (FROZEN=0: the working tree)
R=$GRAFT_REPO_ROOT
T=$R; [ "${FROZEN:-1}" = 1 ] && [ -d $R/ckpt/frozen/blurry-edges_amd/lib ] && T=$R/ckpt/frozen
echo "code tree: $T"
cd $T/blurry-edges_amd
STAGE=${STAGE:-local}
df -h /tmp | tail -n 1; free -g | sed -n 2p
RUN=${RUN:-r05_converged}    # round 6: RUN=r06_converged (the HEAD generator, the balanced training launches)
D=/tmp/be_conv; mkdir -p $D $R/gpurun_out/$RUN
O=$R/gpurun_out/$RUN
NT=${NT:-8000}; NV=${NV:-2000}
t() { date +%s.%N; }
if [ $STAGE != eval ] && [ ! -f $D/data/images_ny_val.npy ]; then
  T0=$(t); python -m be_hip.datagen --data_path $D/data --num_sample_train $NT --num_sample_val $NV > $O/${STAGE}_1_datagen.log 2>&1
  echo "datagen $NT + $NV: $(python -c "print(f'{$(t) - $T0:.1f}')") s" | tee -a $O/${STAGE}_times.txt
fi
mkdir -p $D/w $D/data
case $STAGE in
local)
  EL=${LOCAL_EPOCHS:-1000}; DYN=${LOCAL_DYN:-200}
  T1=$(t)
  python -m be_hip.workflow local_train --data_path $D/data/patches --model_path $D/w --log_path $D/logs --epoch_num $EL --dynamic_epoch $DYN > $O/local_train.log 2>&1
  echo "local_train $EL epochs: $(python -c "print(f'{$(t) - $T1:.1f}')") s" | tee -a $O/${STAGE}_times.txt
  cp $D/logs/exp_local_stage_training.txt $O/local_train_epochs.txt
  cp $D/logs/loss_curve_exp_local_stage.npy $O/
  cp $D/w/best_run_exp_local_stage.pth $O/pretrained_local_stage.pth
  tail -n 4 $O/local_train_epochs.txt
  ;;
global)
  cp $R/ckpt/pretrained_local_stage.pth $D/w/
  EG=${GLOBAL_EPOCHS:-40}
  T2=$(t)
  if [ ! -f $D/data/params_src_val.npy ]; then
    python -m be_hip.workflow global_pre --data_path $D/data --model_path $D/w > $O/global_pre.log 2>&1
    echo "global_pre: $(python -c "print(f'{$(t) - $T2:.1f}')") s" | tee -a $O/${STAGE}_times.txt
  fi
  # the checkpoints live under gpurun_out/ (merged back even when the call is cut at its limit): weights + resume state, ~25 MB
  mkdir -p $O/gw
  [ -f $R/ckpt/global_resume.ckpt ] && cp $R/ckpt/global_resume.ckpt $O/gw/
  T3=$(t)
  python -m be_hip.workflow global_train --data_path $D/data --model_path $O/gw --log_path $D/logs --epoch_num $EG \
      ${GLOBAL_DYN:+--dynamic_epoch $GLOBAL_DYN} --resume --time_budget ${GLOBAL_BUDGET:-780} >> $O/global_train.log 2>&1
  echo "global_train (to epoch budget): $(python -c "print(f'{$(t) - $T3:.1f}')") s" | tee -a $O/${STAGE}_times.txt
  cp $D/logs/exp_global_stage_training.txt $O/global_train_epochs_$(date +%s).txt
  cp $O/gw/best_run_exp_global_stage.pth $O/pretrained_global_stage.pth
  cp $O/gw/global_resume.ckpt $O/global_resume.ckpt
  tail -n 3 $D/logs/exp_global_stage_training.txt
  ;;
eval)
  cp $R/ckpt/pretrained_local_stage.pth $R/ckpt/pretrained_global_stage.pth $D/w/
  cp $D/w/pretrained_global_stage.pth $D/w/pretrained_global_stage_w.pth
  python $T/tests/converged_eval.py --data $D/data --weights $D/w --out $O ${EVAL_N:+--n $EVAL_N} ${ORACLE_N:+--oracle-n $ORACLE_N} ${EVAL_BIG:+--big $EVAL_BIG} 2>&1 | tee $O/eval.log
  ;;
esac
