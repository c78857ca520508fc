#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_exp6.log
{
timeout 1200 python -m pytest tests/test_gpu_f16mx_model.py tests/test_gpu_kernels.py tests/test_fused_micro_steps.py tests/test_reference_ddim_goldens.py tests/test_gpu_entrypoint.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
python - <<'P'
import torch
from ddpo_amd import lib as L
from ddpo_amd.diffusers_patch.scheduling_ddim import DDIMScheduler
dev = "cuda"
sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1)
st = sched.set_timesteps(sched.create_state(device=dev), 50)
consts = sched.kernel_consts(st, 1.0)
for Bt, hw in ((32, 64), (2, 64), (32, 96), (64, 64), (128, 64)):
    chw = 4 * hw * hw
    g = torch.Generator().manual_seed(0)
    ec, eu, x, x2 = (torch.randn(Bt, 4, hw, hw, generator=g).to(dev) for _ in range(4))
    ts = torch.full((Bt,), 481, dtype=torch.int32, device=dev)
    old, adv = torch.full((Bt,), -1.0, device=dev), torch.randn(Bt, generator=g).to(dev)
    pre = L.ddim_logprob_ppo_fwd_bwd(ec, eu, x, x2, ts, old, adv, 5.0, 1e-4, True, consts, group=2)
    fn = lambda: L.ddim_logprob_ppo_fwd_bwd(ec, eu, x, x2, ts, old, adv, 5.0, 1e-4, True, consts, group=2, out=pre)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(100): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 300 * 1e3
    print(f"ppo_fwd_bwd_grouped (graph replay) B={Bt} chw={chw}: {us:.2f} us per launch, {10 * 4 * chw * Bt / us / 1e3:.0f} GB/s algorithmic = {10 * 4 * chw * Bt / us / 1e3 / 8000:.2f} of 8 TB/s")
P
} > $L 2>&1
tail -12 $L
