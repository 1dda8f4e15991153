"""Which PyTorch-ROCm fp32 library path loses precision on gfx950?  Runs the drop-in module's autograd (training) path
against the reference golden g3b under different backend switches.  Diagnostic; prints one line per setting."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import rel_l2
from unet1d import UNet1DConditionModel
from ns2vc_amd.weights import hash_normal, procedural_state_dict

gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
m = UNet1DConditionModel(in_channels=356, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                         cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text", resnet_time_scale_shift="scale_shift")
m.load_state_dict({k: torch.from_numpy(v) for k, v in procedural_state_dict(seed=0).items()}, strict=True)
m = m.cuda().train()
B, T, Lp = 2, 37, 21
x = torch.from_numpy(hash_normal("g3b.x", (B, 100, T))).cuda()
c = torch.from_numpy(hash_normal("g3b.content", (B, 256, T))).cuda()
p = torch.from_numpy(hash_normal("g3b.prompt", (B, Lp, 256))).cuda()
mask = (torch.arange(21)[None, :] < torch.tensor([21, 13])[:, None]).cuda()
t = torch.tensor([499.50003, 499.50003]).cuda()

def run(tag):
    y = m(torch.cat([x, c], dim=1), t, p, encoder_attention_mask=mask).sample
    print(f"{tag:50s} rel_l2 vs reference golden {rel_l2(y.detach().cpu().numpy(), gold['g3b.y']):.3e}", flush=True)

print("defaults: cudnn.allow_tf32", torch.backends.cudnn.allow_tf32, "matmul.allow_tf32", torch.backends.cuda.matmul.allow_tf32,
      "float32_matmul_precision", torch.get_float32_matmul_precision())
run("default (cold)"); run("default (warm)")
torch.backends.cudnn.allow_tf32 = False; run("cudnn.allow_tf32=False")
torch.backends.cudnn.allow_tf32 = True
from torch.nn.attention import sdpa_kernel, SDPBackend
with sdpa_kernel([SDPBackend.MATH]): run("sdpa MATH only")
with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]):
    try: run("sdpa EFFICIENT only")
    except Exception as e: print("sdpa EFFICIENT only: ", type(e).__name__, str(e)[:100])
torch.backends.cudnn.enabled = False; run("cudnn(MIOpen).enabled=False")
torch.backends.cudnn.enabled = True
with sdpa_kernel([SDPBackend.MATH]):
    torch.backends.cudnn.enabled = False; run("MATH sdpa + MIOpen off")

# ---- the conditioning front end (ns2vc_amd.frontend) against golden g10, fused SDPA vs MATH
import json
from ns2vc_amd.frontend import PreModel
from util import procedural_params
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
keys = json.load(open(os.path.join(ROOT, "tests", "golden", "pre_model_state_keys.json")))
cfg = {"phoneme_encoder": {"in_channels": 256, "hidden_channels": 256, "out_channels": 256, "n_layers": 6},
       "prompt_encoder": {"in_channels": 100, "hidden_channels": 256, "out_channels": 256, "n_layers": 6}}
pm = PreModel(cfg).eval(); pm.load_state_dict(procedural_params(keys["keys"], "pre"), strict=True); pm = pm.cuda()
B, T, Lp = 2, 65, 40
lengths, rlens = torch.from_numpy(g["g10.lengths"]).cuda(), torch.from_numpy(g["g10.refer_lengths"]).cuda()
cc = torch.from_numpy(hash_normal("g10.c", (B, 256, T))).cuda() * (torch.arange(T).cuda()[None, None, :] < lengths[:, None, None])
rr = torch.from_numpy(hash_normal("g10.refer", (B, 100, Lp))).cuda() * (torch.arange(Lp).cuda()[None, None, :] < rlens[:, None, None])
def runf(tag):
    content, prompt, _ = pm.infer(cc, rr, lengths, rlens)
    print(f"frontend {tag:40s} content {rel_l2(content.cpu().numpy(), g['g10.content']):.3e} prompt {rel_l2(prompt.cpu().numpy(), g['g10.prompt']):.3e}", flush=True)
torch.backends.cudnn.enabled = True
runf("default")
with sdpa_kernel([SDPBackend.MATH]): runf("sdpa MATH only")
