# End-to-end demo on one MI355X: generate a small basic-shapes set, train the local stage, pre-compute the global inputs,
# train the global stage briefly, evaluate depth on held-out synthetic pairs.  Scaled down from the reference's
# 8000 images / 1000 + 350 epochs so that it finishes in minutes; writes gpurun_out/demo/*.log
set -e
cd $GRAFT_REPO_ROOT/blurry-edges_amd
D=/tmp/be_demo; rm -rf $D; mkdir -p $D ../gpurun_out/demo
O=../gpurun_out/demo
t() { date +%s.%N; }
# sizes: the defaults finish in ~4 minutes; DEMO_* raise them (e.g. DEMO_TRAIN=4000 DEMO_LOCAL_EPOCHS=300 DEMO_GLOBAL_EPOCHS=25: ~15 minutes)
NT=${DEMO_TRAIN:-2000}; NV=${DEMO_VAL:-200}; EL=${DEMO_LOCAL_EPOCHS:-120}; EG=${DEMO_GLOBAL_EPOCHS:-12}
# DEMO_GLOBAL_DYN="a b c": the three epochs of the gamma schedule (default 30 100 200: the depth term only gets its weight late)
T0=$(t); python -m be_hip.datagen --data_path $D/data --num_sample_train $NT --num_sample_val $NV > $O/1_datagen.log 2>&1
T1=$(t); python -m be_hip.workflow local_train --data_path $D/data/patches --model_path $D/w --log_path $D/logs --epoch_num $EL --dynamic_epoch $((EL / 2)) > $O/2_local_train.log 2>&1
cp $D/logs/exp_local_stage_training.txt $O/2_local_train_epochs.txt
cp $D/w/best_run_exp_local_stage.pth $D/w/pretrained_local_stage.pth
T2=$(t); python -m be_hip.workflow global_pre --data_path $D/data --model_path $D/w > $O/3_global_pre.log 2>&1
T3=$(t); python -m be_hip.workflow global_train --data_path $D/data --model_path $D/w --log_path $D/logs --epoch_num $EG ${DEMO_GLOBAL_DYN:+--dynamic_epoch $DEMO_GLOBAL_DYN} > $O/4_global_train.log 2>&1
cp $D/logs/exp_global_stage_training.txt $O/4_global_train_epochs.txt
cp $D/w/best_run_exp_global_stage.pth $D/w/pretrained_global_stage.pth
cp $D/w/best_run_exp_global_stage.pth $D/w/pretrained_global_stage_w.pth   # the name blurry_edges_test.py:187-188 loads for --densify w
T4=$(t); mkdir -p $D/test
python - <<PY
import numpy as np
d = "$D"
for src, dst in (("images_ny_val", "images_ny"), ("image_depths_val", "depth_maps"), ("alphas_val", "alphas")):
    np.save(f"{d}/test/{dst}.npy", np.load(f"{d}/data/{src}.npy")[:50])
PY
python -m be_hip.workflow eval --data_path $D/test --model_path $D/w --densify w > $O/5_eval.log 2>&1
T5=$(t)
python - <<PY > $O/summary.txt
t = [float(x) for x in "$T0 $T1 $T2 $T3 $T4 $T5".split()]
names = ["datagen ($NT + $NV image pairs)", "local_train ($EL epochs, batch 64)", "global_pre ($NT + $NV image pairs)",
         "global_train ($EG epochs of $NT / 8 steps, batch 8)", "eval (50 pairs, 147x147)"]
for n, a, b in zip(names, t[:-1], t[1:]):
    print(f"{n:55s} {b - a:8.1f} s")
PY
cat $O/summary.txt; tail -n 3 $O/5_eval.log; head -n 3 $O/2_local_train.log | cut -c1-100; tail -n 2 $O/2_local_train.log; tail -n 2 $O/4_global_train.log
