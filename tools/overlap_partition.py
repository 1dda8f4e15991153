#!/usr/bin/env python3
"""r5, round-4 review item 4: a CU partition of one MI355X between the denoiser and the PyTorch front / back end.

    python tools/overlap_partition.py > gpurun_out/overlap_partition.txt

1. learns how the bits of a hipExtStreamCreateWithCUMask mask map onto (XCD, CU) -- one single-bit stream per bit, a resident grid on it,
   HW_REG_XCC_ID / HW_REG_HW_ID of its blocks (ns2vc_debug_placement);
2. times 32 x 10 s batches through the three-stage pipeline (front end | 20-step UniPC denoiser | vocoder): sequential, overlapped on three
   plain streams (the r3 arrangement), and overlapped with the PyTorch stages on k CUs of every XCD and the denoiser on the rest
   (k = 2, 4, 6), plus the denoiser alone on the reduced CU set (what the partition costs it)."""
import ctypes as C
import functools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import procedural_params                      # noqa: E402
from ns2vc_amd import _lib                               # noqa: E402
from ns2vc_amd._lib import check                         # noqa: E402
from ns2vc_amd.engine import Stream                      # noqa: E402
from ns2vc_amd.frontend import PreModel                 # noqa: E402
from ns2vc_amd.pipeline import Denoiser, OverlappedPipeline   # noqa: E402
from ns2vc_amd.vocoder import VocosDecoder              # noqa: E402
from ns2vc_amd.weights import procedural_state_dict     # noqa: E402

PRE_CFG = {"phoneme_encoder": {"in_channels": 256, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2},
           "prompt_encoder": {"in_channels": 100, "hidden_channels": 256, "out_channels": 256, "n_layers": 6, "p_dropout": 0.2}}


print = functools.partial(print, flush=True)


def placement(stream, n=512, spin=200):
    lib = _lib.load()
    out = (C.c_uint32 * (2 * n))()
    check(lib.ns2vc_debug_placement(stream.ptr if stream is not None else None, n, spin, out), "placement")
    a = np.frombuffer(out, dtype=np.uint32).reshape(n, 2)
    return a[:, 0].copy(), a[:, 1].copy()


def main():
    dev = torch.device("cuda", 0)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    # ---- 1. the mask's bit order (tools/cu_mask_probe.py, profiles/r05_cu_mask_probe.txt): bit b = CU b // 8 of XCD b % 8 -- every contiguous run of
    # 32 bits gives each XCD four CUs; an XCD left without any enabled CU runs the stream's kernels UNMASKED, so a partition must leave
    # every XCD some CUs on both sides.  The dispatcher keeps dealing workgroup ids round robin over all eight XCDs under any such mask.
    by_xcd = {k: [b for b in range(ncu) if b % 8 == k] for k in range(8)}
    x, _ = placement(None, n=1024, spin=200)
    print("# unmasked stream, block i -> XCC:", x[:16].tolist(), " round robin:", bool(all(x[i] == x[i % 8] for i in range(1024))))

    # ---- 2. the pipeline
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "pre_model_state_keys.json")))
    pre = PreModel(PRE_CFG).eval()
    pre.load_state_dict(procedural_params(keys["keys"], "pre"), strict=True)
    pre = pre.to(dev)
    torch.manual_seed(0)
    voc = VocosDecoder().eval().to(dev)
    den = Denoiser(procedural_state_dict(seed=0), precision_check=None)
    B, T, Lp, steps, n = 32, 938, 469, 20, 3
    g = torch.Generator(device=dev).manual_seed(5)
    c = torch.randn((B, 256, T), device=dev, generator=g)
    refer = torch.randn((B, 100, Lp), device=dev, generator=g)
    lengths, rlens = torch.full((B,), T, device=dev), torch.full((B,), Lp, device=dev)
    noise = torch.randn((B, 100, T), device=dev, generator=g)

    def pre_fn(k):
        content, prompt, mask = pre.infer(c, refer, lengths, rlens)
        return {"content": content, "prompt": prompt, "prompt_mask": mask, "noise": noise}

    def post_fn(latent, k):
        return voc.decode(latent)

    def timed(fn, reps=1):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best / n * 1e3

    def sequential():
        for k in range(n):
            cd = pre_fn(k)
            post_fn(den.sample(cd["content"], cd["prompt"], cd["prompt_mask"], cd["noise"], solver="unipc", steps=steps), k)

    cd0 = pre_fn(0)

    def denoiser_only(stream):
        with torch.cuda.stream(stream):
            for k in range(n):
                den.sample(cd0["content"], cd0["prompt"], cd0["prompt_mask"], cd0["noise"], solver="unipc", steps=steps)

    def stages_only(stream):
        with torch.cuda.stream(stream):
            for k in range(n):
                post_fn(noise, k)
                pre_fn(k)

    sequential()
    rows = [("sequential, one stream", timed(sequential))]
    plain = OverlappedPipeline(den, pre_fn, post_fn, solver="unipc", steps=steps)
    plain.run([0])
    rows.append(("overlapped, three plain streams (r3)", timed(lambda: plain.run(list(range(n))))))
    s0 = torch.cuda.Stream(dev)
    rows.append(("denoiser alone, whole chip", timed(lambda: denoiser_only(s0))))
    rows.append(("front end + vocoder alone, whole chip", timed(lambda: stages_only(s0))))
    for name, ms in rows:
        print(f"{ms:9.2f}  {name}")
    # Under a partition a cooperative grid is no longer one resident round of workgroups: siblings wait for each other in vain (bounded,
    # ~130 us each).  So the partitioned runs use the gn_coop = 0 plan (every column tile builds its rows), the documented switch for it.
    print("# plain runs done; partitioned runs with the gn_coop = 0 plan")
    den.set_option("gn_coop", False)
    sequential()
    rows.append(("sequential, one stream, gn_coop = 0 plan", timed(sequential)))
    print(f"{rows[-1][1]:9.2f}  {rows[-1][0]}")
    for k in (4, 8, 2):
      try:
        stage = [b for x in range(8) for b in by_xcd[x][-k:]]
        rest = [b for b in range(ncu) if b not in set(stage)]
        sd = Stream(cu_mask=rest)
        ss = Stream(cu_mask=stage)
        xd, _ = placement(sd, n=1024, spin=200)
        rr = bool(all(xd[i] == xd[i % 8] for i in range(1024)))
        tsd, tss = torch.cuda.ExternalStream(sd.ptr, device=dev), torch.cuda.ExternalStream(ss.ptr, device=dev)
        denoiser_only(tsd); stages_only(tss)
        t_den, t_st = timed(lambda: denoiser_only(tsd)), timed(lambda: stages_only(tss))
        alone = den.engine.gn_coop_alone(stream=tsd)
        pipe = OverlappedPipeline(den, pre_fn, post_fn, solver="unipc", steps=steps, stage_cus=stage, denoiser_cus=rest)
        pipe.run([0])
        rows.append((f"partition: stages on {len(stage)} CUs ({k} per XCD), denoiser on {len(rest)}: overlapped", timed(lambda: pipe.run(list(range(n))))))
        rows.append((f"   ... denoiser alone on its {len(rest)} CUs (round robin kept: {rr}; prologue workgroups alone: {alone})", t_den))
        rows.append((f"   ... front end + vocoder alone on their {len(stage)} CUs", t_st))
        pipe2 = OverlappedPipeline(den, pre_fn, post_fn, solver="unipc", steps=steps, stage_cus=stage)
        pipe2.run([0])
        rows.append((f"   ... stages on {len(stage)} CUs, denoiser unmasked: overlapped", timed(lambda: pipe2.run(list(range(n))))))
        for name, ms in rows[-4:]:
            print(f"{ms:9.2f}  {name}")
      except Exception as ex:
        print(f"# partition k={k} failed: {ex!r}")
    print(f"# ms per 32 x 10 s batch ({steps}-step UniPC, fp16 engine, fp32 PyTorch stages), one run of {n} batches each")
    for name, ms in rows:
        print(f"{ms:9.2f}  {name}")


if __name__ == "__main__":
    main()
