#!/usr/bin/env python3
"""Benchmark of the NS2VC denoiser hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "denoiser-steps/sec (10s@24kHz Vocos latent, bs32)",
configs[2]): 10 s utterances (T = 938 Vocos frames), batch 32 per GPU, prompt Lp = 469,
UniPC-bh2 order 2, hipGraph-captured loop, 16-bit MFMA operands (fp16: the 16-bit mode
that meets the 1e-3 parity bar; --precision bf16|fp32 for the others).  One "step" = one
denoiser evaluation (UNet forward on the whole batch) + the fused solver update.  A timed
JOB is one sampling run of exactly K steps: the once-per-utterance condition hoisting
(set_condition) + K graph replays + the layout change back to (B,100,T), inputs already
resident in HBM, and for N > 1 the all-gather of finished latents, bracketed by barrier +
synchronize on both sides, MAX over ranks.  After W warm-up steps the job is repeated
--reps times (default 5); `ms_per_step` / `value` are the MEDIAN job (min and all jobs are
in the line too).  Synthetic data (seeded hash), procedural weights of the production
architecture.

N > 1: launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment),
or plainly as `python bench.py --gpus N` -- then this process spawns the N ranks itself
(127.0.0.1 rendezvous) and fails loudly if fewer than N devices are visible.

`python bench.py --dry-dist N` rehearses that multi-rank path on CPU (gloo, stub engine): the driver's 8-GPU run is the first time
RCCL sees more than one rank, but not the first time the harness around it runs.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), with
  roofline      dominant kernel family (implicit GEMM: gemm4 + fused feed-forward + row chains) vs the dense MFMA peak;
                `achieved` = the family's share of the TIMED loop, `isolated` = its launches timed back to back with HIP events
  parity        rel-L2 of the timed precision's predicted latent vs the oracle AT THE BENCH SHAPE (B=32), and
                `sampled_latent`: the timed solver's output (captured loop, B=32) vs oracle.sampler_ref on the first utterances
  fp32_parity_mode   the same job and roofline in the exact-fp32 precision
  other_configs BASELINE configs 2 and 5 (default and --attn-fp8) timed the same way, with parity against the fp32 engine
  strong_scaling     BASELINE config 4's global batch (256) split over the N ranks
  cpu_baseline  the oracle on the host cores: all cores (P processes x T threads) and one process, at B=32
"""
from __future__ import annotations

import argparse
import datetime
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
METRIC = "denoiser-steps/sec (10s@24kHz Vocos latent, bs32)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5, help="timed jobs of --steps steps each; the median is reported")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (weak scaling)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--prompt-frames", type=int, default=469)
    ap.add_argument("--solver", default="unipc", choices=["unipc", "dpmsolver++"])
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--attn-fp8", action="store_true", help="PV product of every attention on the fp8 MFMA (BASELINE config 5's fp8 path; "
                    "costs parity, see DESIGN.md section 4)")
    ap.add_argument("--full", action="store_true", help="also run the long legs (self_check, fp32_parity_mode, bf16_as_stated, other_configs = BASELINE configs 2 and 5, "
                    "strong_scaling, the larger CPU grid); their results go to the detail file, the stdout line stays the compact one "
                    "(printed before the long legs start and again as the last line)")
    ap.add_argument("--detail-json", default=os.path.join(ROOT, "bench_detail.json"), help="where rank 0 writes the full record (every block of the compact line "
                    "unabridged + the --full legs); '' = do not write")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU-baseline / parity legs")
    ap.add_argument("--skip-fp32", action="store_true", help="(--full) skip the fp32_parity_mode block")
    ap.add_argument("--skip-others", action="store_true", help="(--full) skip the other_configs block (BASELINE configs 2 and 5)")
    ap.add_argument("--skip-strong", action="store_true", help="(--full) skip the strong_scaling block (global batch 256 split over the ranks)")
    ap.add_argument("--tail-fp32", type=int, default=0, help="last evaluations of the timed loop on a second, fp32 engine (mixed precision; "
                    "0 = the headline's pure 16-bit loop)")
    ap.add_argument("--strong-batch", type=int, default=256, help="global batch of the strong_scaling block (BASELINE config 4: 256)")
    ap.add_argument("--dry-dist", type=int, default=0, metavar="N", help="harness rehearsal WITHOUT GPUs: N ranks on the gloo backend, the engine replaced by "
                    "a sleep + deterministic fill; exercises the spawn, every barrier, the all-gather of latents, the per-rank reduction, the "
                    "strong_scaling block and the JSON line exactly as --gpus N does (tests/test_cpu.py runs it at N = 2 and 8)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of TIMED CPU work for the baseline grid (split over its legs; the parity "
                    "references are extra and unbudgeted: one batch-B forward + the sampled latent of 2 utterances)")
    ap.add_argument("--ops", default="", help="write the per-launch table (name, kind, ms, GFLOP, MB) to this file")
    ap.add_argument("--detail", action="store_true", help="print the per-kernel-family table to stderr")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)     # internal: one all-core baseline worker
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-frames", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pin", default="", help=argparse.SUPPRESS)                  # internal: "lo-hi" core range of one all-core worker
    ap.add_argument("--cpu-batch", type=int, default=2, help=argparse.SUPPRESS)       # internal: batch of one all-core worker
    ap.add_argument("--cpu-weights", default="", help=argparse.SUPPRESS)              # internal: flat fp32 weight file the workers map (written once by the parent)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline = the oracle (oracle/unet_ref.py, pinned bit-exact to the reference by tests/golden) on the host cores
# ------------------------------------------------------------------------------------------------
def _host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _cpu_quota() -> float | None:
    """CPUs the container may actually burn: the cgroup CFS quota (cpu.max = "<quota> <period>").  The GPU boxes of this pool show
    256 logical CPUs and a quota of 16 (measured r6): every thread beyond the quota only adds throttling, which is why the r1-r5 grids
    got slower with more processes."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            return None if q == "max" else float(q) / float(per)
        except Exception:
            pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def _numa_layout() -> str:
    try:
        out = []
        base = "/sys/devices/system/node"
        for n in sorted(d for d in os.listdir(base) if d.startswith("node") and d[4:].isdigit()):
            out.append(f"{n}={open(os.path.join(base, n, 'cpulist')).read().strip()}")
        return " ".join(out) or "unknown"
    except Exception:
        return "unknown"


def _weights_file(W, path):
    """The oracle's weights as ONE flat fp32 file + an index, written once by the parent: the grid's workers map it (shared page cache)
    instead of each regenerating 66 M parameters inside a CPU quota shared with the timed window."""
    index, off = {}, 0
    with open(path, "wb") as f:
        for k, v in W.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            index[k] = (off, list(a.shape))
            f.write(a.tobytes())
            off += a.size
    with open(path + ".json", "w") as f:
        json.dump(index, f)


def _oracle_setup(threads: int, weights_path: str = ""):
    import torch
    from ns2vc_amd.spec import UNetConfig
    torch.set_num_threads(max(1, threads))
    cfg = UNetConfig()
    if weights_path:
        flat = np.memmap(weights_path, dtype=np.float32, mode="c")       # copy-on-write mapping: read-only in effect, shared between the workers
        index = json.load(open(weights_path + ".json"))
        return cfg, {k: torch.from_numpy(flat[o:o + int(np.prod(sh, dtype=np.int64))].reshape(sh)) for k, (o, sh) in index.items()}
    from ns2vc_amd.weights import procedural_state_dict
    return cfg, {k: torch.from_numpy(v) for k, v in procedural_state_dict(cfg, 0).items()}


def bench_inputs(tag: str, B: int, T: int, Lp: int):
    from ns2vc_amd.spec import UNetConfig
    from ns2vc_amd.weights import hash_normal
    cfg = UNetConfig()
    return (hash_normal(tag + ".noise", (B, cfg.latent_channels, T)), hash_normal(tag + ".content", (B, cfg.content_channels, T)),
            hash_normal(tag + ".prompt", (B, Lp, cfg.cross_attention_dim)))


def cpu_worker(seconds: float, threads: int, T: int, Lp: int, B: int = 2, weights_path: str = ""):
    """One worker of a grid leg: oracle forwards at batch B for `seconds`; prints samples and elapsed time."""
    import torch
    from oracle import unet_ref
    cfg, P = _oracle_setup(threads, weights_path)
    x, content, prompt = (torch.from_numpy(a) for a in bench_inputs(f"cpuw{os.getpid() % 97}", B, T, Lp))
    sample = torch.cat([x, content], dim=1)
    mask = torch.ones(B, Lp, dtype=torch.bool)
    t = torch.full((B,), 500.0)
    unet_ref.unet_forward(P, cfg, sample, t, prompt, mask)            # warm-up (also faults the weight pages in)
    print("READY", flush=True)
    sys.stdin.readline()                                              # all workers of a leg start together
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        unet_ref.unet_forward(P, cfg, sample, t, prompt, mask)
        n += B
    print(json.dumps({"samples": n, "seconds": time.perf_counter() - t0}), flush=True)


def cpu_grid(cores_eff: int, full: bool):
    """(processes, threads, batch) legs of the baseline grid.  Many small processes is what a throughput-minded host would run (r5 review):
    with E usable cores the default legs are E/8 x 8, E/4 x 4 and E/2 x 2 threads at batch 2 (= 32x8, 64x4, 128x2 on an unthrottled 256-CPU
    host; 2x8, 4x4, 8x2 under this pool's 16-CPU quota); --full adds E x 1 at batch 1 and E/16 x 16 at batch 4."""
    legs = [(max(1, cores_eff // t), t, 2) for t in (8, 4, 2) if t <= cores_eff]
    if full:
        legs += [(cores_eff, 1, 1)] + ([(cores_eff // 16, 16, 4)] if cores_eff >= 32 else [])
    return [g for g in dict.fromkeys(legs) if g[0] >= 1]


def cpu_baseline(T: int, Lp: int, B: int, budget_s: float, sampler=None, W=None, full=False):
    """(1) one process at the bench batch (its output is also the parity reference, timed as the `single_process` leg); (2) the grid of
    cpu_grid(): P processes x T threads, every worker pinned to its own disjoint cores, weights mapped from one shared file, started
    together after their warm-up, samples/s summed, the best leg (or the single process, if faster) is `value`; (3) `sampler` = (solver,
    steps, items): oracle.sampler_ref on the first `items` utterances of the same inputs = the reference of parity.sampled_latent.
    Cores: what the cgroup lets the container burn (cpu.max), not what lscpu shows."""
    import torch
    from oracle import unet_ref
    visible, quota = _host_cores(), _cpu_quota()
    eff = max(1, min(visible, int(quota + 0.5))) if quota else visible
    th = min(eff, 32)
    wpath = ""
    if W is not None:
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        wpath = os.path.join(shm, f"ns2vc_oracle_w_{os.getpid()}.f32")
        _weights_file(W, wpath)
    try:
        cfg, P = _oracle_setup(th, wpath)
        x, content, prompt = (torch.from_numpy(a) for a in bench_inputs("bench.r0", B, T, Lp))
        mask = torch.ones(B, Lp, dtype=torch.bool)
        t_par = torch.linspace(40.0, 960.0, B)
        sample = torch.cat([x, content], dim=1)
        unet_ref.unet_forward(P, cfg, sample[:2], t_par[:2], prompt[:2], mask[:2])     # warm-up
        t0 = time.perf_counter()
        y_ref = unet_ref.unet_forward(P, cfg, sample, t_par, prompt, mask)
        dt = time.perf_counter() - t0
        single = {"sample_steps_per_s": B / dt, "processes": 1, "threads_per_process": th, "batch_per_process": B, "seconds": dt}
        # ---- the grid
        sweep, agg = [], None
        grid = cpu_grid(eff, full)
        secs = max(2.5, budget_s / max(len(grid), 1))
        avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(visible))
        for nproc, tha, bw in grid:
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(secs), "--cpu-threads", str(tha), "--cpu-frames", str(T),
                   "--prompt-frames", str(Lp), "--cpu-batch", str(bw)] + (["--cpu-weights", wpath] if wpath else [])
            env = dict(os.environ, OMP_NUM_THREADS=str(tha), MKL_NUM_THREADS=str(tha), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
            procs = [subprocess.Popen(cmd + ["--cpu-pin", f"{i * tha}-{(i + 1) * tha}"], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL, text=True) for i in range(nproc)]
            leg = {"processes": nproc, "threads_per_process": tha, "batch_per_process": bw}
            try:
                for p in procs:
                    if "READY" not in p.stdout.readline():
                        raise RuntimeError("cpu worker failed to start")
                for p in procs:
                    p.stdin.write("go\n"); p.stdin.flush()
                outs = [json.loads(p.stdout.readline()) for p in procs]
                leg.update(sample_steps_per_s=sum(o["samples"] / o["seconds"] for o in outs), seconds=secs)
            except Exception as ex:
                leg["error"] = repr(ex)
            finally:
                for p in procs:
                    try:
                        p.kill()
                    except Exception:
                        pass
            sweep.append(leg)
            if "sample_steps_per_s" in leg and (agg is None or leg["sample_steps_per_s"] > agg["sample_steps_per_s"]):
                agg = leg
        use = agg if agg and agg["sample_steps_per_s"] > single["sample_steps_per_s"] else single
        used = min(eff, use["processes"] * use["threads_per_process"])
        # ---- the sampled latent of the timed solver on the first utterances (the oracle treats utterances independently)
        samp = None
        if sampler is not None:
            from oracle import sampler_ref
            solver, steps, nb = sampler
            nb = max(1, min(nb, B))
            torch.set_num_threads(th)
            tc, tp, tm = content[:nb], prompt[:nb], mask[:nb]

            def x0(xx, tt):
                return unet_ref.denoiser(P, cfg, xx, tc, tp, tm, tt)
            t0 = time.perf_counter()
            betas = sampler_ref.linear_betas(1000)
            ys = (sampler_ref.unipc_bh2(x0, betas, x[:nb].clone(), steps) if solver == "unipc" else
                  sampler_ref.dpm_solver_pp_2m(x0, betas, x[:nb].clone(), steps, 2 if steps >= 2 else 1))
            samp = {"y": ys.numpy(), "items": nb, "seconds": time.perf_counter() - t0}
        pinned = "workers pinned to disjoint contiguous logical CPUs (sched_setaffinity), weights mapped from one shared file"
        out = {"value": use["sample_steps_per_s"] / B, "unit": f"denoiser-steps/s (batch {B})", "cores": used, "host_cores": visible,
               "cgroup_cpu_quota": quota, "kind": "port", "sample_steps_per_s": use["sample_steps_per_s"],
               "sample": (f"oracle UNet forward (torch CPU fp32), T={T}, Lp={Lp}: best of {len(sweep) + 1} legs = {use['processes']} proc x {use['threads_per_process']} thr x "
                          f"batch {use['batch_per_process']}, {use['seconds']:.1f} s window; {visible} CPUs visible, cgroup quota {quota if quota else 'none'}"),
               "numa": _numa_layout(), "pinning": pinned, "single_process": single, "all_cores": agg, "all_cores_sweep": sweep}
        return out, (x.numpy(), content.numpy(), prompt.numpy(), mask.numpy(), t_par.numpy(), y_ref.numpy()), samp
    finally:
        for fpath in (wpath, wpath + ".json"):
            if wpath and os.path.exists(fpath):
                try:
                    os.remove(fpath)
                except OSError:
                    pass


# ------------------------------------------------------------------------------------------------
def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on
    127.0.0.1) and relay rank 0's JSON line."""
    if not a.dry_dist:
        from ns2vc_amd import engine as E
        n_dev = E.device_count()
        if n_dev < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {n_dev} ROCm device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NS2VC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench.py: rank return codes {rcs}")
    return 0


def family_table(eng, stream, ops_path=""):
    ops = eng.op_info(0)
    eng.profile_forward(reps=2, stream=stream)        # warm-up pass
    ms = eng.profile_forward(reps=8, stream=stream)   # 8 back-to-back launches per HIP-event pair, on the launch stream
    names = {0: "other", 1: "implicit_gemm", 2: "attention", 3: "norm_stats", 4: "copy"}
    fam = {}
    if ops_path:
        with open(ops_path, "w") as f:
            for (name, kind, fl, by), m in zip(ops, ms):
                f.write(f"{name}\t{names[kind]}\t{m*1e3:.1f}us\t{fl/1e9:.3f}GF\t{by/1e6:.2f}MB\t{(fl/(m*1e-3)/1e12 if m > 0 else 0):.1f}TF/s\t{(by/(m*1e-3)/1e9 if m > 0 else 0):.0f}GB/s\n")
    for (name, kind, fl, by), m in zip(ops, ms):
        f = fam.setdefault(names[kind], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        f["launches"] += 1; f["ms"] += float(m); f["flops"] += fl; f["bytes"] += by
    return fam


def roofline_block(fam, precision, step_ms, gflop_sample, B, shape):
    """Dominant kernel family = implicit GEMM (conv3ts_kernel + gemm4_kernel + the fused feed-forward ffn_kernel + geglu_kernel + rowchain_kernel).
    `achieved` is what the family reaches INSIDE the timed loop: its algorithmic FLOP / (step time x the family's share of the
    step), the share taken from the per-launch HIP-event timings (rocprofv3 cannot run inside bench.py; its kernel-trace
    figure for the same command is stamped as `rocprof` when profiles/ holds one for this shape and precision).  `isolated`
    is the same family with every launch repeated 8x back to back between one event pair (L2-warm, no neighbours): an upper
    bound on the kernels themselves, ~8 % above the in-loop figure."""
    peak = MFMA_PEAK_TFLOPS[precision]
    g = fam.get("implicit_gemm", {"launches": 1, "ms": 1.0, "flops": 0.0, "bytes": 0.0})
    iso_tflops = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    total_iso_ms = sum(v["ms"] for v in fam.values())
    share = g["ms"] / total_iso_ms if total_iso_ms > 0 else 0.0
    loop_ms = step_ms * share
    gemm_tflops = g["flops"] / (loop_ms * 1e-3) / 1e12 if loop_ms > 0 else 0.0
    # HBM bytes per launch of the same family and the rocprofv3 kernel-trace time of the family: rocprofv3 cannot run inside
    # bench.py, so both come from the committed passes of this very command (profiles/rNN_pmc_hbm_traffic.json: FETCH_SIZE
    # x2-corrected + WRITE_SIZE; profiles/rNN_family_times.json), stamped with the commit they were measured at and quoted
    # only for the workload / precision they were measured on
    traffic = traffic_src = traffic_commit = traffic_cmd = None
    rocprof = None
    pdir = os.path.join(ROOT, "profiles")
    names = sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []
    for fn in (f for f in names if f.endswith("_pmc_hbm_traffic.json")):
        try:
            with open(os.path.join(pdir, fn)) as fh:
                tj = json.load(fh)
        except Exception:
            continue
        if tj.get("precision", "bf16") == precision and tuple(tj.get("shape", (32, 938, 469))) == tuple(shape) and tj.get("families", {}).get("implicit_gemm"):
            traffic = tj["families"]["implicit_gemm"]["hbm_mb_per_launch"] * 1e6
            traffic_src = f"profiles/{fn} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bytes per launch)"
            traffic_commit = tj.get("commit")
            traffic_cmd = tj.get("source")
            break
    for fn in (f for f in names if f.endswith("_family_times.json")):
        try:
            with open(os.path.join(pdir, fn)) as fh:
                fj = json.load(fh)
        except Exception:
            continue
        if fj.get("precision") == precision and tuple(fj.get("shape", ())) == tuple(shape) and fj.get("families", {}).get("implicit_gemm"):
            fm = fj["families"]["implicit_gemm"]["ms_per_step"]
            rocprof = {"family_ms_per_step": fm, "tflops": g["flops"] / (fm * 1e-3) / 1e12 if fm > 0 else 0.0,
                       "frac": (g["flops"] / (fm * 1e-3) / 1e12 / peak) if fm > 0 else 0.0, "step_ms_under_rocprof": fj.get("step_ms"),
                       "source": f"profiles/{fn}", "command": fj.get("source"), "measured_at": fj.get("commit")}
            break
    return {
        "bound": "mfma", "kernel": "implicit-GEMM family: conv3ts_kernel (conv1d k3, tap-sharing) + gemm4_kernel (stride-2 / upsample convs, conv1d k1, linear) + ffn_kernel (fused feed-forward) + geglu_kernel (token-stationary GEGLU projection, dim 384) + rowchain_kernel (token-local linear chains)",
        "achieved": gemm_tflops, "peak": peak, "unit": "TFLOP/s", "frac": gemm_tflops / peak,
        "achieved_method": "family FLOP / (timed ms_per_step x the family's share of the per-launch HIP-event times)",
        "family_share_of_step": share, "family_ms_in_loop": loop_ms,
        # (the two cross-checks as flat scalars too: a consumer that keeps only scalar roofline keys still sees them)
        "frac_isolated": iso_tflops / peak, "frac_rocprof": rocprof["frac"] if rocprof else None,
        "family_ms_rocprof": rocprof["family_ms_per_step"] if rocprof else None, "rocprof_measured_at": rocprof["measured_at"] if rocprof else None,
        "isolated": {"tflops": iso_tflops, "frac": iso_tflops / peak, "ms_per_step": g["ms"],
                     "method": "every launch 8x back to back between one HIP event pair on the launch stream (L2-warm)"},
        "rocprof": rocprof,
        "traffic": traffic, "traffic_source": traffic_src, "traffic_command": traffic_cmd, "traffic_measured_at": traffic_commit,
        "algorithmic_bytes_per_launch": g["bytes"] / max(g["launches"], 1),
        "launches_per_step": g["launches"], "avg_launch_us": loop_ms * 1e3 / max(g["launches"], 1),
        "algorithmic_gflop_per_launch": g["flops"] / 1e9 / max(g["launches"], 1),
        "algorithmic_hbm_gbs": g["bytes"] / (loop_ms * 1e-3) / 1e9 if loop_ms > 0 else 0.0,
        "whole_step": {"algorithmic_tflop_per_step": gflop_sample * B / 1e3, "ms_per_step": step_ms,
                       "achieved_tflops": gflop_sample * B / 1e3 / (step_ms * 1e-3), "frac_of_mfma_peak": gflop_sample * B / 1e3 / (step_ms * 1e-3) / peak},
        "families": {k: {"launches": v["launches"], "ms_per_step_isolated": round(v["ms"], 4),
                         "ms_per_step_in_loop": round(step_ms * v["ms"] / total_iso_ms, 4) if total_iso_ms > 0 else None,
                         "tflops_isolated": (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0),
                         "algorithmic_gbs_isolated": (v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0)} for k, v in fam.items()},
    }


def compact_line(d: dict) -> dict:
    """The ONE stdout line (driver contract + roofline + cpu_baseline + parity), kept under 4 KB: the r5 line had grown to 21.5 KB and the
    driver could no longer parse it.  Everything else is in the detail file (--detail-json)."""
    def rnd(v, n=6):
        return round(v, n) if isinstance(v, float) else v
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    o = {k: rnd(d[k]) for k in keep}
    o["data"] = "synthetic" if d["data"].startswith("synthetic") else d["data"]
    o["config"] = d["config"]
    r = d.get("roofline")
    if r:
        o["roofline"] = {"bound": r["bound"], "kernel": "implicit-GEMM family: conv3ts + gemm4 + ffn + geglu + rowchain kernels (all MFMA launches but attention)",
                         "achieved": rnd(r["achieved"], 2), "peak": r["peak"], "unit": r["unit"], "frac": rnd(r["frac"], 4),
                         "frac_isolated": rnd(r["frac_isolated"], 4), "frac_rocprof": rnd(r["frac_rocprof"], 4), "rocprof_measured_at": r["rocprof_measured_at"],
                         "traffic": r["traffic"], "traffic_source": (r["traffic_source"] or "").split(" ")[0] or None, "traffic_measured_at": r["traffic_measured_at"],
                         "algorithmic_bytes_per_launch": rnd(r["algorithmic_bytes_per_launch"], 0), "algorithmic_gflop_per_launch": rnd(r["algorithmic_gflop_per_launch"], 4),
                         "launches_per_step": r["launches_per_step"], "avg_launch_us": rnd(r["avg_launch_us"], 3), "family_ms_in_loop": rnd(r["family_ms_in_loop"], 4),
                         "whole_step_frac_of_mfma_peak": rnd(r["whole_step"]["frac_of_mfma_peak"], 4)}
    else:
        o["roofline"] = None
    c = d.get("cpu_baseline")
    o["cpu_baseline"] = ({k: rnd(c.get(k), 5) for k in ("value", "unit", "cores", "host_cores", "cgroup_cpu_quota", "kind", "sample")} if c else None)
    if d.get("speedup_vs_cpu_baseline"):
        o["speedup_vs_cpu_baseline"] = rnd(d["speedup_vs_cpu_baseline"], 1)
    pz = d.get("parity")
    if pz:
        sl = pz.get("sampled_latent")
        o["parity"] = {"rel_l2_vs_oracle": rnd(pz["rel_l2_vs_oracle"], 8), "sampled_latent": rnd(sl["rel_l2_vs_oracle"], 8) if sl else None,
                       "sampled_items": sl["items"] if sl else None, "tolerance": pz["tolerance"], "mode": pz["mode"],
                       "reference": "oracle/unet_ref.py + oracle/sampler_ref.py (pinned to the reference by tests/golden), B=32 forward / timed loop"}
    else:
        o["parity"] = None
    o["launches_per_step"] = d["launches_per_step"]
    o["rccl_ranks"] = d["rccl_ranks"]
    o["finite"] = d["finite"]
    o["graph_equals_eager"] = (d["loop_check"] or {}).get("graph_loop_equals_eager_loop")
    o["jobs_ms"] = [rnd(v, 3) for v in d["timing"]["jobs_ms"]]
    o["per_rank_ms_per_step"] = [rnd(v, 4) for v in d["per_rank_ms_per_step"]]
    for k in ("all_gather_ms", "all_gather_bytes", "spawned_by", "dry_dist"):
        if k in d:
            o[k] = rnd(d[k], 4)
    o["device"] = d["device"]
    o["detail"] = "bench_detail.json (every block unabridged; --full adds fp32 / bf16 / configs 2 and 5 / strong scaling)"
    return o


def write_detail(path: str, d: dict) -> None:
    if not path:
        return
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(d, f, indent=1)
        os.replace(tmp, path)
    except Exception as ex:                              # a read-only tree must not take the line down
        print(f"bench.py: could not write {path}: {ex!r}", file=sys.stderr)


class DryEngine:
    """--dry-dist: stands in for ns2vc_amd.engine.Engine with the same call sequence.  A job sleeps ~0.2 ms per step and utterance
    and returns x_T / 2 (deterministic, so the gathered latents can be checked on every rank)."""

    def __init__(self, B, K):
        self.B, self.K = B, K

    def set_condition(self, *a, **k):
        pass

    def sample(self, x, use_graph=True, stream=None, tail=None, tail_steps=0):
        time.sleep(2e-4 * self.K * max(self.B, 1))
        x.mul_(0.5)

    def launches(self):
        return (0, 0)

    def workspace_bytes(self):
        return 0

    def attn_fallbacks(self, reset=True, stream=None):
        return 0

    def close(self):
        pass


class _DryStream:
    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _DryEvent:
    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_ms(self, other):
        return (other.t - self.t) * 1e3


def main():
    a = parse()
    if a.dry_dist:
        a.gpus = a.dry_dist
    T_frames = int(math.floor(24000 * a.seconds / 256)) + 1
    if a.cpu_worker > 0:
        if a.cpu_pin:                                   # one all-core worker on its own disjoint set of cores (before torch starts its thread pool)
            try:
                lo, hi = (int(v) for v in a.cpu_pin.split("-"))
                avail = sorted(os.sched_getaffinity(0))
                os.sched_setaffinity(0, set(avail[lo:hi]))
            except Exception:
                pass
        cpu_worker(a.cpu_worker, a.cpu_threads or 8, a.cpu_frames or T_frames, a.prompt_frames, max(1, a.cpu_batch), a.cpu_weights)
        return
    if "RANK" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from ns2vc_amd import engine as E
    from ns2vc_amd.spec import PUBLISHED_GFLOP, UNetConfig, algorithmic_gflop_per_sample_step, frames_for_seconds
    from ns2vc_amd.weights import procedural_state_dict
    from ns2vc_amd.dist import gather_latents

    dry = bool(a.dry_dist)
    if dry:
        torch.set_num_threads(1)
        if world > 1:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available() or E.device_count() == 0:
            raise SystemExit("bench.py needs an MI355X: no ROCm device visible (there is no CPU fallback)")
        if E.device_count() <= local:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {E.device_count()} ROCm device(s) visible")
        torch.cuda.set_device(local)
        E.set_device(local)
        if world > 1:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        dev = torch.device("cuda", local)

    cfg = UNetConfig()
    B, T, Lp, K = a.batch, frames_for_seconds(a.seconds), a.prompt_frames, a.steps
    assert T == T_frames
    order = 2 if K >= 2 else 1
    solver = a.solver
    W = None if dry else procedural_state_dict(cfg, 0)
    use_graph = not a.no_graph
    stream = _DryStream() if dry else torch.cuda.Stream(device=dev)
    on_stream = (lambda: stream) if dry else (lambda: torch.cuda.stream(stream))
    new_event = _DryEvent if dry else E.Event
    dev_sync = (lambda: None) if dry else (lambda: torch.cuda.synchronize(dev))

    def build(precision, B_=None, T_=None, solver_=None, K_=None, attn_fp8=None):
        B_, T_, solver_, K_ = B_ if B_ is not None else B, T_ or T, solver_ or solver, K_ or K
        if dry:
            return DryEngine(B_, K_)
        eng = E.Engine(cfg, precision=precision)
        if (a.attn_fp8 if attn_fp8 is None else attn_fp8) and precision != "fp32":
            eng.set_option("attn_fp8", True)
        eng.load_state_dict(W)
        eng.prepare(B_, T_, Lp)
        eng.load_sampler(solver_, K_, order=2 if K_ >= 2 else 1)
        return eng

    def dev_inputs(tag, B_, T_):
        if dry:                                               # only the latent matters to the harness
            g = torch.Generator().manual_seed(1234 + sum(map(ord, tag)))
            n = torch.randn((B_, cfg.latent_channels, T_), generator=g)
            return {"noise": n, "content": None, "prompt": None, "mask": None, "x": torch.empty_like(n)}
        n_np, c_np, p_np = bench_inputs(tag, B_, T_, Lp)
        c, p_, n = (torch.from_numpy(v).to(dev) for v in (c_np, p_np, n_np))
        return {"noise": n, "content": c, "prompt": p_, "mask": torch.ones((B_, Lp), dtype=torch.uint8, device=dev), "x": torch.empty_like(n)}

    def barrier():
        dev_sync()
        if world > 1:
            dist.barrier()
        dev_sync()

    def timed_jobs(eng, io, K_, warmup_steps, reps, with_gather, tail=None, tail_steps=0, n_total=None):
        """`reps` timed jobs of exactly K_ steps, each bracketed by barrier + synchronize; returns per-job wall seconds
        (MAX over ranks), GPU-event ms of the last job and the all-gather seconds of the last job.  With `tail`, the last
        `tail_steps` evaluations run on that (fp32) engine: both engines' condition hoisting is inside the timed job."""
        x = io["x"]

        def job():
            x.copy_(io["noise"])                             # x_T
            eng.set_condition(io["content"], io["prompt"], io["mask"], stream=stream)
            if tail is not None:
                tail.set_condition(io["content"], io["prompt"], io["mask"], stream=stream)
            eng.sample(x, use_graph=use_graph, stream=stream, tail=tail, tail_steps=tail_steps)
        walls, gpu_ms, t_gather = [], 0.0, 0.0
        with on_stream():
            for _ in range(max(1, math.ceil(warmup_steps / max(K_, 1)))):
                job()
            stream.synchronize()
            for _ in range(reps):
                barrier()
                ev0, ev1 = new_event(), new_event()
                t0 = time.perf_counter()
                ev0.record(stream)
                job()
                ev1.record(stream)
                if with_gather:
                    stream.synchronize()
                    tg = time.perf_counter()
                    full = gather_latents(x, n_total)
                    dev_sync()
                    t_gather = time.perf_counter() - tg
                    assert full.shape[0] == n_total
                    if dry:                                   # the stub engine halves x_T: every rank must hold every rank's slice, in order
                        lo_, hi_ = io.get("range", (0, 0))
                        assert torch.equal(full[lo_:hi_], x) and bool(torch.isfinite(full).all())
                stream.synchronize()
                barrier()
                wall = time.perf_counter() - t0
                if world > 1:
                    tt = torch.tensor([wall], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    wall = float(tt.item())
                walls.append(wall)
                gpu_ms = ev0.elapsed_ms(ev1)
        return walls, gpu_ms, t_gather

    def rel_l2_dev(y, ref):
        return float((y.double() - ref.double()).norm() / ref.double().norm())

    io = dev_inputs(f"bench.r{rank}", B, T)
    io["range"] = (rank * B, (rank + 1) * B)
    x = io["x"]
    eng = build(a.precision)
    tail_eng = build("fp32") if (a.tail_fp32 > 0 and a.precision != "fp32") else None
    launches, workspace_gb = eng.launches()[0], eng.workspace_bytes() / 1e9
    reps = max(1, a.reps)
    walls, gpu_ms, t_gather = timed_jobs(eng, io, K, a.warmup, reps, world > 1, tail_eng, a.tail_fp32, B * world)
    wall = statistics.median(walls)
    finite = bool(torch.isfinite(x).all().item())
    x_timed = x.clone()
    # the timed (captured-graph) loop must return exactly what the same loop launched eagerly returns
    loop_check = None
    attn_fb = None if dry else eng.attn_fallbacks(reset=True, stream=stream)          # workgroups that paid an attention kernel twice, all jobs so far
    gn_alone = None if dry else eng.gn_coop_alone(reset=True, stream=stream)         # cooperative GroupNorm prologue: workgroups that waited in vain for a sibling, all jobs so far
    if use_graph and not dry:
        with torch.cuda.stream(stream):
            x.copy_(io["noise"])
            eng.set_condition(io["content"], io["prompt"], io["mask"], stream=stream)
            if tail_eng is not None:
                tail_eng.set_condition(io["content"], io["prompt"], io["mask"], stream=stream)
            eng.sample(x, use_graph=False, stream=stream, tail=tail_eng, tail_steps=a.tail_fp32)
            stream.synchronize()
            loop_check = {"graph_loop_equals_eager_loop": bool(torch.equal(x, x_timed)), "steps": K}
    per_rank_ms = [wall * 1e3 / K]
    gather_ms = 0.0
    if world > 1:
        mine = torch.tensor([statistics.median(walls) * 1e3 / K, t_gather * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(v[0]) for v in allr]
        gather_ms = max(float(v[1]) for v in allr)

    out = line = None
    ref = samp = None
    gflop_sample = PUBLISHED_GFLOP.get((T, Lp), algorithmic_gflop_per_sample_step(T, Lp))
    def parity_of(engine, precision):
        xr, cr, pr, mr, tr, yr = ref
        d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (xr, cr, pr, mr.astype(np.uint8), tr.astype(np.float32))]
        out = torch.empty_like(d[0])
        with torch.cuda.stream(stream):
            engine.set_condition(d[1], d[2], d[3], stream=stream)
            engine.forward(d[0], d[4], out, stream=stream)
            stream.synchronize()
        y = out.cpu().numpy().astype(np.float64)
        return {"mode": precision, "rel_l2_vs_oracle": float(np.linalg.norm(y - yr) / np.linalg.norm(yr)),
                "shape": {"batch": B, "frames": T, "prompt_frames": Lp}, "tolerance": 1e-3,
                "reference": "oracle/unet_ref.py (pinned bit-exact to the reference by tests/golden), one UNet forward, per-item timesteps 40..960"}

    if rank == 0:
        step_ms = wall * 1e3 / K
        fam = roof = None
        if not dry:
            fam = family_table(eng, stream, a.ops)
            roof = roofline_block(fam, a.precision, step_ms, gflop_sample, B, (B, T, Lp))

        # ---- CPU baseline (oracle on the host cores) + parity of the timed precision at the bench shape
        cpu = parity = None
        if world == 1 and not a.skip_cpu and not dry:
            try:
                cpu, ref, samp = cpu_baseline(T, Lp, B, a.cpu_budget, sampler=(solver, K, 2), W=W, full=a.full)
            except Exception as ex:                      # the baseline leg must never take the GPU number down
                cpu = {"value": None, "unit": f"denoiser-steps/s (batch {B})", "cores": _host_cores(), "kind": "port", "sample": f"failed: {ex!r}"}
        if ref is not None:
            parity = parity_of(eng, a.precision)
            if samp is not None:                       # the TIMED loop's output (same inputs: bench.r0) on the utterances the oracle sampled
                nb = samp["items"]
                ys = x_timed[:nb].cpu().numpy().astype(np.float64)
                parity["sampled_latent"] = {
                    "rel_l2_vs_oracle": float(np.linalg.norm(ys - samp["y"]) / np.linalg.norm(samp["y"])), "items": nb, "of_batch": B,
                    "solver": solver, "steps": K, "tail_fp32": a.tail_fp32, "oracle_seconds": round(samp["seconds"], 1),
                    "reference": f"oracle/sampler_ref.py (pinned to the reference's own samplers by tests/golden) on utterances 0..{nb - 1} of the timed batch, identical noise"}

        value = world * K / wall
        out = {
            "metric": METRIC, "value": value, "unit": "denoiser-steps/s (batch 32 per GPU, whole job)",
            "n_gpus": world, "steps": K, "warmup": a.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic (seeded hash inputs, procedural weights of the production UNet1DConditionModel)",
            "config": {"workload": f"{a.seconds:g} s utterance (T={T} Vocos frames), batch {B}/GPU, prompt Lp={Lp}, {K}-step {solver} order {order}, "
                                   f"{'hipGraph-captured' if use_graph else 'eager'} loop, {a.precision} MFMA operands{' + fp8 PV in attention' if a.attn_fp8 else ''}"
                                   f"{f' + last {a.tail_fp32} evaluations fp32' if a.tail_fp32 else ''}; timed job = set_condition + {K} steps"
                                   + (" + all-gather of latents" if world > 1 else ""),
                       "global_batch": B * world, "frames": T, "prompt_frames": Lp, "solver": solver, "parallelism": f"dp{world}"},
            "timing": {"jobs": reps, "statistic": "median", "jobs_ms": [w * 1e3 for w in walls], "min_ms_per_step": min(walls) * 1e3 / K,
                       "max_ms_per_step": max(walls) * 1e3 / K},
            "sample_steps_per_s": value * B, "rtf": wall / (B * a.seconds), "gpu_event_ms": gpu_ms, "finite": finite, "loop_check": loop_check,
            "launches_per_step": launches, "workspace_gb": workspace_gb, "device": "none (dry run)" if dry else E.device_info(),
            "attention_fallback_workgroups": attn_fb,
            "gn_prologue_workgroups_alone": gn_alone,
            "xcd_round_robin": (None if dry else E.xcd_round_robin()),
            "rccl_ranks": (dist.get_world_size() if (world > 1 and dist.is_initialized()) else 0), "per_rank_ms_per_step": per_rank_ms,
            "roofline": roof, "parity": parity, "cpu_baseline": cpu,
            "self_check": None, "fp32_parity_mode": None, "bf16_as_stated": None, "other_configs": None, "strong_scaling": None,
        }
        if dry:
            out["dry_dist"] = {"backend": "gloo", "engine": "stub (sleep + deterministic fill)", "note": "harness rehearsal, not a measurement"}
            out["data"] = "dry run"
        if world > 1:
            out["all_gather_ms"] = gather_ms
            out["all_gather_bytes"] = int(B * world * cfg.latent_channels * T * 4)     # what every rank receives: the finished latents of the global batch
            out["spawned_by"] = "bench.py" if os.environ.get("NS2VC_BENCH_SPAWNED") else "launcher"
        if cpu and cpu.get("value"):
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        if a.detail and roof:
            for k, v in roof["families"].items():
                print(f"  {k:14s} {v}", file=sys.stderr)
        # ---- THE line: compact (< 4 KB), printed as soon as the headline, its roofline, parity and CPU baseline exist -- before any long leg
        cl = compact_line(out)
        line = json.dumps(cl)
        for k in ("detail", "jobs_ms", "device", "per_rank_ms_per_step"):      # (never let the size guard take the line down: shed optional keys instead)
            if len(line) < 4096:
                break
            cl.pop(k, None)
            line = json.dumps(cl)
        print(line, flush=True)
        write_detail(a.detail_json, out)

    # ---- the long legs (--full): results go to the detail file only ---------------------------------------------------------------------
    full1 = bool(a.full and rank == 0 and world == 1 and not dry)
    if full1:
        t_full = time.perf_counter()
        # ---- what the served API measures about itself on this workload: Denoiser's precision self-check (16-bit vs exact-fp32 engine at
        # the first / middle / last evaluation point of the trajectory, batch figure and worst utterance) and the LayerNorm guard's ratio
        self_check = None
        if full1 and a.precision != "fp32" and not a.skip_fp32:
            try:
                from ns2vc_amd.pipeline import Denoiser
                den = Denoiser(W, cfg, precision=a.precision)
                with torch.cuda.stream(stream):
                    ys = den.sample(io["content"], io["prompt"], io["mask"].bool(), io["noise"], solver=solver, steps=K, order=order, use_graph=use_graph,
                                    tail_fp32=a.tail_fp32)
                    stream.synchronize()
                self_check = {
                    "what": "ns2vc_amd.pipeline.Denoiser on the timed inputs: relative L2 of the 16-bit engine vs the exact-fp32 engine at the first / middle / "
                            "last evaluation point of the sampling trajectory (gate: both figures <= threshold, else the Denoiser serves from fp32)",
                    "points": [{"t": t_, "batch_rel_l2": b_, "worst_utterance_rel_l2": w_} for t_, b_, w_ in den.precision_errors],
                    "precision_error_seen": den.precision_error_seen, "precision_error_worst_item": den.precision_error_worst_item,
                    "threshold": den.precision_check, "serving_fp32": den.serving_fp32,
                    "ln_ratio_seen": den.ln_ratio_seen, "ln_guard": den.ln_guard,
                    "denoiser_output_equals_timed_loop": bool(torch.equal(ys, x_timed))}
                del den
            except Exception as ex:
                self_check = {"error": repr(ex)}

        # ---- the exact-fp32 precision: same job, same roofline definition (peak = 157.3 TFLOP/s fp32 MFMA)
        fp32_block = None
        bf16_block = None
        e32 = None
        if full1 and a.precision != "fp32" and not a.skip_fp32:
            eng.close()
            e32 = build("fp32")
            w32, _, _ = timed_jobs(e32, io, K, K, min(reps, 3), False)
            x32 = x.clone()
            ms32 = statistics.median(w32) * 1e3 / K
            fam32 = family_table(e32, stream)
            fp32_block = {"dtype": "fp32", "ms_per_step": ms32, "value": K / statistics.median(w32), "jobs_ms": [w * 1e3 for w in w32],
                          "roofline": roofline_block(fam32, "fp32", ms32, gflop_sample, B, (B, T, Lp)),
                          "sampled_latent_vs_timed_precision": rel_l2_dev(x_timed, x32)}
            if ref is not None:
                fp32_block["parity"] = parity_of(e32, "fp32")
            # BASELINE configs[2] literally says "bf16": the headline workload with bf16 operands (same MFMA rate and bytes as the fp16 that is
            # served and timed above; bf16 cannot meet the 1e-3 tolerance -- BASELINE.md section 2), with its parity beside it (r5)
            if a.precision == "fp16":
                try:
                    eb = build("bf16")
                    wb, _, _ = timed_jobs(eb, io, K, K, min(reps, 3), False)
                    xb = x.clone()
                    bf16_block = {"dtype": "bf16", "what": "the headline job (set_condition + 20 UniPC steps, batch 32, 10 s) with bf16 MFMA operands, as BASELINE configs[2] names it",
                                  "ms_per_step": statistics.median(wb) * 1e3 / K, "value": K / statistics.median(wb), "jobs_ms": [w * 1e3 for w in wb],
                                  "launches_per_step": eb.launches()[0], "sampled_latent_vs_fp32_loop": rel_l2_dev(xb, x32),
                                  "fp16_sampled_latent_vs_fp32_loop": rel_l2_dev(x_timed, x32), "tolerance": 1e-3}
                    if ref is not None:
                        bf16_block["parity"] = parity_of(eb, "bf16")
                    eb.close()
                except Exception as ex:
                    bf16_block = {"error": repr(ex)}
            # the mixed-precision loop at the headline shape: the last two evaluations on this fp32 engine
            if a.tail_fp32 == 0:
                try:
                    em = build(a.precision)
                    wmx, _, _ = timed_jobs(em, io, K, K, min(reps, 3), False, e32, 2)
                    fp32_block["mixed_precision_tail2"] = {
                        "what": f"{a.precision} for the first {K - 2} evaluations, fp32 for the last 2 (ns2vc_sampler_handoff); both engines' set_condition inside the job",
                        "ms_per_step": statistics.median(wmx) * 1e3 / K, "value": K / statistics.median(wmx),
                        "sampled_latent_vs_fp32_loop": rel_l2_dev(x, x32), "pure_16bit_vs_fp32_loop": rel_l2_dev(x_timed, x32)}
                    if samp is not None:
                        ys = x[:samp["items"]].cpu().numpy().astype(np.float64)
                        fp32_block["mixed_precision_tail2"]["sampled_latent_vs_oracle"] = float(np.linalg.norm(ys - samp["y"]) / np.linalg.norm(samp["y"]))
                    em.close()
                except Exception as ex:
                    fp32_block["mixed_precision_tail2"] = {"error": repr(ex)}
            e32.close()

        # ---- BASELINE configs 2 and 5, timed the same way (one GPU): parity against the exact-fp32 engine at the same shape
        others = None
        if full1 and not a.skip_others:
            others = []
            try:
                eng.close()
            except Exception:
                pass
            specs = [("configs[1]: 10 s utterance, 50-step DPM-Solver, batch 8", 10.0, 8, "dpmsolver++", 50, False),
                     ("configs[4] shape, 16-bit attention: 30 s utterance, 50-step DPM-Solver, batch 8", 30.0, 8, "dpmsolver++", 50, False),
                     ("configs[4] as stated (fp8 MFMA attention path): 30 s utterance, 50-step DPM-Solver, batch 8", 30.0, 8, "dpmsolver++", 50, True)]
            ref32 = {}
            for name, secs, B2, solver2, K2, fp8 in specs:
                try:
                    T2 = frames_for_seconds(secs)
                    io2 = dev_inputs(f"other.{secs:g}", B2, T2)
                    if (T2, B2) not in ref32:          # exact-fp32 engine: the forward and the whole loop as references
                        r32 = build("fp32", B_=B2, T_=T2, solver_=solver2, K_=K2)
                        t_par = torch.linspace(40.0, 960.0, B2, device=dev)
                        y32 = torch.empty_like(io2["noise"])
                        with torch.cuda.stream(stream):
                            r32.set_condition(io2["content"], io2["prompt"], io2["mask"], stream=stream)
                            r32.forward(io2["noise"], t_par, y32, stream=stream)
                            io2["x"].copy_(io2["noise"])
                            r32.sample(io2["x"], use_graph=use_graph, stream=stream)
                            stream.synchronize()
                        ref32[(T2, B2)] = (r32, t_par, y32, io2["x"].clone())
                    r32, t_par, y32, s32 = ref32[(T2, B2)]
                    e2 = build(a.precision, B_=B2, T_=T2, solver_=solver2, K_=K2, attn_fp8=fp8)
                    tail2 = 2 if a.precision != "fp32" else 0
                    w2, _, _ = timed_jobs(e2, io2, K2, K2, 3, False, r32 if tail2 else None, tail2)
                    s_mixed = io2["x"].clone()
                    w2p, _, _ = timed_jobs(e2, io2, K2, K2, 3, False)
                    s_pure = io2["x"].clone()
                    y2 = torch.empty_like(io2["noise"])
                    with torch.cuda.stream(stream):
                        e2.set_condition(io2["content"], io2["prompt"], io2["mask"], stream=stream)
                        e2.forward(io2["noise"], t_par, y2, stream=stream)
                        stream.synchronize()
                    oracle_fwd = None
                    if not a.skip_cpu:                 # one utterance through the ORACLE at this shape (CPU, ~seconds): parity vs the reference itself, not vs the fp32 engine
                        from oracle import unet_ref
                        if "P" not in ref32:
                            ref32["P"] = _oracle_setup(min(_host_cores(), 16))[1]
                        t0o = time.perf_counter()
                        yo = unet_ref.denoiser(ref32["P"], cfg, io2["noise"][:1].cpu(), io2["content"][:1].cpu(), io2["prompt"][:1].cpu(),
                                               torch.ones(1, Lp, dtype=torch.bool), t_par[:1].cpu()).double()
                        oracle_fwd = {"utterance": 0, "rel_l2_vs_oracle": float((y2[:1].cpu().double() - yo).norm() / yo.norm()),
                                      "fp32_engine_rel_l2_vs_oracle": float((y32[:1].cpu().double() - yo).norm() / yo.norm()),
                                      "oracle_seconds": round(time.perf_counter() - t0o, 1),
                                      "reference": "oracle/unet_ref.py, one UNet forward of utterance 0 at this shape (timestep 40)"}
                    g2 = PUBLISHED_GFLOP.get((T2, Lp), algorithmic_gflop_per_sample_step(T2, Lp))
                    ms2, ms2p = statistics.median(w2) * 1e3 / K2, statistics.median(w2p) * 1e3 / K2
                    fam2 = family_table(e2, stream)
                    others.append({
                        "config": name, "workload": f"{secs:g} s (T={T2}), batch {B2}, Lp={Lp}, {K2}-step {solver2} order 2, captured loop, {a.precision} operands"
                                                    + (", PV of every attention on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)" if fp8 else "")
                                                    + (f", last {tail2} evaluations on the fp32 engine" if tail2 else ""),
                        "dtype": a.precision + ("+fp8 PV" if fp8 else ""), "tail_fp32": tail2, "ms_per_step": ms2, "value": K2 / statistics.median(w2),
                        "unit": f"denoiser-steps/s (batch {B2}, whole job)", "sample_steps_per_s": B2 * K2 / statistics.median(w2),
                        "rtf": statistics.median(w2) / (B2 * secs), "jobs_ms": [w * 1e3 for w in w2],
                        "pure_16bit_loop": {"ms_per_step": ms2p, "value": K2 / statistics.median(w2p)},
                        "parity": {"reference": "the exact-fp32 engine at the same shape and inputs (itself 1e-6 from the oracle: tests/test_engine_gpu.py, fp32_parity_mode)",
                                   "forward_rel_l2": rel_l2_dev(y2, y32), "sampled_latent_rel_l2": rel_l2_dev(s_mixed, s32),
                                   "sampled_latent_rel_l2_pure_16bit": rel_l2_dev(s_pure, s32), "tolerance": 1e-3, "forward_vs_oracle": oracle_fwd},
                        "roofline": roofline_block(fam2, a.precision, ms2p, g2, B2, (B2, T2, Lp)),
                        "launches_per_step": e2.launches()[0]})
                    e2.close()
                except Exception as ex:
                    others.append({"config": name, "error": repr(ex)})
            for k32, v32 in ref32.items():
                if k32 != "P":
                    v32[0].close()

        out.update(self_check=self_check, fp32_parity_mode=fp32_block, bf16_as_stated=bf16_block, other_configs=others,
                   full_legs_seconds=round(time.perf_counter() - t_full, 1))
        write_detail(a.detail_json, out)
    # ---- strong scaling: BASELINE config 4's global batch split over the ranks (every rank takes part; rank 0 reports).  It runs
    # AFTER everything the headline needs (the headline engines are closed by now), and the ranks AGREE on having built their shard
    # engine before anyone enters the timed jobs' collectives: a rank that fails (e.g. out of memory) cannot leave the others in a barrier.
    strong = None
    if a.full and not a.skip_strong:
        from ns2vc_amd.dist import shard_range
        GB = a.strong_batch
        lo, hi = shard_range(GB, rank, world)
        es, ios, err = None, None, None
        try:
            eng.close()
            if tail_eng is not None:
                tail_eng.close()
            es = build(a.precision, B_=hi - lo)
            ios = dev_inputs(f"strong.r{rank}", hi - lo, T)
            ios["range"] = (lo, hi)
        except Exception as ex:
            err = repr(ex)
        ok = 0 if err else 1
        if world > 1:
            tt = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            ok = int(tt.item())
        if ok:
            # never take the headline down (round-4 advice): a failure inside the timed jobs becomes an error entry of the one JSON line.
            # (A rank that raises in here leaves the others in a collective; they are released by the process group's timeout, and the
            # second agreement below turns "somebody failed" into an error entry on every rank instead of a half-reported number.)
            try:
                ws, _, tg = timed_jobs(es, ios, K, K, min(reps, 3), world > 1, None, 0, GB)
                wm = statistics.median(ws)
                strong = {"global_batch": GB, "per_rank_batch": hi - lo, "n_gpus": world, "steps": K, "solver": solver, "ms_per_step": wm * 1e3 / K,
                          "value": K / wm, "unit": f"denoiser-steps/s at global batch {GB} (strong scaling: the batch is split over the ranks)",
                          "sample_steps_per_s": GB * K / wm, "jobs_ms": [w * 1e3 for w in ws], "all_gather_ms": tg * 1e3 if world > 1 else 0.0,
                          "scaling": "strong"}
            except Exception as ex:
                strong = {"error": repr(ex), "global_batch": GB}
            if world > 1:
                try:
                    tt = torch.tensor([0 if "error" in strong else 1], device=dev, dtype=torch.int32)
                    dist.all_reduce(tt, op=dist.ReduceOp.MIN)
                    if int(tt.item()) == 0 and "error" not in strong:
                        strong = {"error": "another rank failed inside the strong-scaling jobs", "global_batch": GB}
                except Exception as ex:
                    strong = {"error": "agreement after the strong-scaling jobs failed: " + repr(ex), "global_batch": GB}
        else:
            strong = {"error": err or "another rank could not build its shard engine", "global_batch": GB}
        if es is not None:
            es.close()
        del ios


    if rank == 0 and a.full:
        out["strong_scaling"] = strong
        write_detail(a.detail_json, out)
        print(line, flush=True)                     # the same compact line again, as the LAST line of stdout
    for e_ in (eng, tail_eng):
        try:
            if e_ is not None:
                e_.close()
        except Exception:
            pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
