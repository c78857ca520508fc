#!/bin/bash
# round 6: the two long parity records of VERDICT r05 next 6c / 6d — the full-size train step at the bench's 16 fused micro-steps against the oracle's 16
# accumulated steps, and the full-scale entrypoint (BASELINE configs[1] = C2: real pipeline/policy_gradient.py, 3 epochs, jpeg reward, reward vs wall-clock)
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r06_parity_train_fuse16.log
DDPO_PARITY_LOG=gpurun_out/r06_parity_train_fuse16.log DDPO_TRAIN_PARITY_FUSE=${FUSE:-16} timeout 3000 python -m pytest tests/test_gpu_train_parity.py -k sd15_full_size -m gpu -q -x -p no:cacheprovider -s > gpurun_out/r06_pytest_train_fuse16.log 2>&1
tail -5 gpurun_out/r06_pytest_train_fuse16.log
rm -rf /tmp/e2e_r06; mkdir -p /tmp/e2e_r06
DDPO_ALLOW_SYNTHETIC=1 timeout 1500 python pipeline/policy_gradient.py --dataset compressed_animals --num_train_epochs 3 --save_freq 1000 --logbase /tmp/e2e_r06/run > gpurun_out/r06_e2e_entrypoint_full_scale.log 2>&1
echo "pg exit $?" >> gpurun_out/r06_e2e_entrypoint_full_scale.log
python - >> gpurun_out/r06_e2e_entrypoint_full_scale.log 2>&1 <<'P'
import glob, numpy as np
for f in sorted(glob.glob('/tmp/e2e_r06/run/**/reward_vs_wallclock.npy', recursive=True)):
    a = np.load(f)
    print('reward_vs_wallclock.npy', f, 'shape', a.shape)
    print(a)
P
tail -12 gpurun_out/r06_e2e_entrypoint_full_scale.log
