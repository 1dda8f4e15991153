#!/usr/bin/env python3
"""The three-stream pipeline (front end | denoiser | vocoder) on 32 x 10 s batches, for a rocprofv3 kernel trace:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -- python $REPO/tools/overlap_run.py [--sequential]
tools/overlap_analyze.py turns the trace into how much of the PyTorch stages' kernel time ran CONCURRENTLY with denoiser kernels."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import procedural_params                      # noqa: E402
from ns2vc_amd.frontend import PreModel                 # noqa: E402
from ns2vc_amd.pipeline import Denoiser, OverlappedPipeline   # noqa: E402
from ns2vc_amd.vocoder import VocosDecoder              # noqa: E402
from ns2vc_amd.weights import procedural_state_dict     # noqa: E402

PRE_CFG = {"phoneme_encoder": {"in_channels": 256, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2},
           "prompt_encoder": {"in_channels": 100, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2}}


def main():
    sequential = "--sequential" in sys.argv
    ac = torch.float16 if "--autocast" in sys.argv else None
    dev = torch.device("cuda", 0)
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "pre_model_state_keys.json")))
    pre = PreModel(PRE_CFG).eval()
    pre.load_state_dict(procedural_params(keys["keys"], "pre"), strict=True)
    pre = pre.to(dev)
    torch.manual_seed(0)
    voc = VocosDecoder().eval().to(dev)
    den = Denoiser(procedural_state_dict(seed=0), precision_check=None)
    B, T, Lp, steps, n = 32, 938, 469, 20, 4
    g = torch.Generator(device=dev).manual_seed(5)
    c = torch.randn((B, 256, T), device=dev, generator=g)
    refer = torch.randn((B, 100, Lp), device=dev, generator=g)
    lengths, rlens = torch.full((B,), T, device=dev), torch.full((B,), Lp, device=dev)
    noise = torch.randn((B, 100, T), device=dev, generator=g)

    def pre_fn(k):
        content, prompt, mask = pre.infer(c, refer, lengths, rlens, autocast=ac)
        return {"content": content, "prompt": prompt, "prompt_mask": mask, "noise": noise}

    def post_fn(latent, k):
        return voc.decode(latent, autocast=ac)

    pipe = OverlappedPipeline(den, pre_fn, post_fn, solver="unipc", steps=steps)
    pipe.run([0])                                       # warm-up: plan, graph capture, library handles
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if sequential:
        for k in range(n):
            cd = pre_fn(k)
            post_fn(den.sample(cd["content"], cd["prompt"], cd["prompt_mask"], cd["noise"], solver="unipc", steps=steps), k)
    else:
        pipe.run(list(range(n)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": "sequential" if sequential else "overlapped", "autocast": bool(ac), "batches": n, "ms_per_batch": dt / n * 1e3,
                      "rtf": dt / (n * B * T * 256 / 24000.0)}))


if __name__ == "__main__":
    main()
