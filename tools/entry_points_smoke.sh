#!/bin/bash
# Every kept entry point once, end to end, on the GPU box (small networks, synthetic data): the regression train / test mains, the
# projector's with the REFERENCE's own argv (GenProjector/train_laval.sh, test.sh) + --synthetic, the joint main, the two-rank
# refusal of --gpu_ids 0,1 in one process.  Prints one OK / FAILED line per entry point.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
CK=$(mktemp -d /tmp/eml_ck.XXXX)
run() { name=$1; shift; if timeout 600 "$@" > $CK/$name.log 2>&1; then echo "OK      $name"; else echo "FAILED  $name (rc=$?)"; tail -5 $CK/$name.log; fi; }
run regression_train python -m emlight_amd.RegressionNetwork.train --synthetic --batch_size 2 --anchors 32 --crop_hw 64 96 --max_iters 3 --save_dir $CK/reg --summary_dir $CK/sum
run regression_test python -m emlight_amd.RegressionNetwork.test --synthetic --anchors 32 --crop_hw 64 96 --max_images 2 --checkpoint $CK/reg/latest_net.pth --results_dir $CK/res
LAVAL="--name lavalindoor --dataset_mode lavalindoor --dataroot /home/fangneng.zfn/datasets/LavalIndoor/tpami/ --display_freq 1000 --batchSize 2 --niter 100 --niter_decay 100"
run projector_train python -m emlight_amd.GenProjector.train $LAVAL --gpu_ids 0 --synthetic --ngf 8 --ndf 8 --max_iters 2 --checkpoints_dir $CK/ck
run projector_train_resume python -m emlight_amd.GenProjector.train $LAVAL --gpu_ids 0 --continue_train --synthetic --ngf 8 --ndf 8 --max_iters 1 --checkpoints_dir $CK/ck
run projector_test python -m emlight_amd.GenProjector.test --name lavalindoor --checkpoints_dir $CK/ck --which_epoch latest --dataset_mode lavalindoor --dataroot /home/fangneng.zfn/datasets/LavalIndoor/test/ --synthetic --ngf 8 --how_many 1 --results_dir $CK/out
if python -m emlight_amd.GenProjector.train $LAVAL --gpu_ids 0,1 --synthetic > $CK/refuse.log 2>&1; then echo "FAILED  gpu_ids 0,1 in one process was accepted"; else grep -q "torchrun" $CK/refuse.log && echo "OK      projector_train refuses --gpu_ids 0,1 in one process, names the torchrun launch"; fi
run joint python -m emlight_amd.joint --batch 2 --anchors 32 --crop_hw 64 96 --ngf 8 --ndf 8 --max_iters 2
rm -rf $CK
