#!/bin/bash
# round 3: the entrypoints at full SD-1.x scale with synthetic weights — policy gradient (2 epochs at the defaults) and the RWR pair
# (sample 16 images into a local shard store, then one fine-tuning epoch of 4 steps).  gpurun --timeout 900 -- 'bash tools/r03_e2e.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp DDPO_ALLOW_SYNTHETIC=1
rm -rf /tmp/e2e_r03
timeout 300 python pipeline/policy_gradient.py --dataset compressed-animals --num_train_epochs 2 --save_freq 1000 --logbase /tmp/e2e_r03/run > gpurun_out/r03_e2e_entrypoint_full_scale.log 2>&1; echo "pg exit $?" >> gpurun_out/r03_e2e_entrypoint_full_scale.log
grep -E "images in|train steps in|mean reward|exit" gpurun_out/r03_e2e_entrypoint_full_scale.log | tail -8
timeout 300 python pipeline/sample.py --dataset compressed_animals_rwr --logbase /tmp/e2e_r03/rwr --max_samples 16 --n_samples_per_device 8 --local_size 16 > gpurun_out/r03_e2e_rwr_full_scale.log 2>&1; echo "sample exit $?" >> gpurun_out/r03_e2e_rwr_full_scale.log
timeout 300 python pipeline/finetune.py --dataset compressed_animals_rwr --logbase /tmp/e2e_r03/rwr --num_train_epochs 1 --train_batch_size 4 --save_freq 1000 >> gpurun_out/r03_e2e_rwr_full_scale.log 2>&1; echo "finetune exit $?" >> gpurun_out/r03_e2e_rwr_full_scale.log
grep -E "exit|loss|images|epoch|sample" gpurun_out/r03_e2e_rwr_full_scale.log | tail -12 | cut -c1-200
