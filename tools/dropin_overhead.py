"""per-call cost of the zero-change drop-in path (module.forward on a fresh torch.cat every step, eager launches, content re-staged
per call) beside the captured loop of ns2vc_amd.pipeline.Denoiser, 10 s x batch 32"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unet1d import UNet1DConditionModel
from ns2vc_amd.pipeline import Denoiser
from ns2vc_amd.weights import procedural_state_dict
dev = torch.device("cuda", 0)
B, T, Lp, K = 32, 938, 469, 20
W = {k: torch.from_numpy(v) for k, v in procedural_state_dict(seed=0).items()}
m = UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8, cross_attention_dim=256,
                         attention_head_dim=8, addition_embed_type="text", resnet_time_scale_shift="scale_shift", engine_precision="fp16")
m.load_state_dict(W, strict=True); m = m.to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((B, 100, T), device=dev, generator=g); c = torch.randn((B, 256, T), device=dev, generator=g)
p = torch.randn((B, Lp, 256), device=dev, generator=g); mask = torch.ones((B, Lp), dtype=torch.bool, device=dev)
def loop():
    xx = x
    with torch.no_grad():
        for k in range(K):
            t = torch.full((B,), 999.0 - 50.0 * k, device=dev)
            xx = m(torch.cat([xx, c], dim=1), t, p, encoder_attention_mask=mask).sample      # the reference's call (model.py:411-415)
    return xx
loop(); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(); torch.cuda.synchronize(); t_mod = (time.perf_counter() - t0) / K
den = Denoiser(procedural_state_dict(seed=0))
den.sample(c, p, mask, x, solver="unipc", steps=K); torch.cuda.synchronize()
t0 = time.perf_counter(); den.sample(c, p, mask, x, solver="unipc", steps=K); torch.cuda.synchronize(); t_den = (time.perf_counter() - t0) / K
print(f"drop-in module.forward (eager, per call): {t_mod * 1e3:.2f} ms; Denoiser captured loop: {t_den * 1e3:.2f} ms/step; prompt hoists {m.prompt_hoists} in {m.engine_calls} calls")
