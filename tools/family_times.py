#!/usr/bin/env python3
"""rocprofv3 kernel-trace view of the bench loop per kernel family: profiles/<TAG>_family_times.json, which bench.py stamps
into roofline.rocprof (the in-loop figure it reports itself comes from HIP events; this is the profiler's).
usage: family_times.py TAG kernel_stats.csv bench_line.json OUTDIR
  kernel_stats.csv = `rocprofv3 --kernel-trace --stats` of `python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong ...`,
  bench_line.json  = the JSON line that run printed (launches per step and the shape come from it)."""
import csv, json, os, re, subprocess, sys

tag, f_stats, f_bench, outdir = sys.argv[1:5]


def short(name):
    m = re.search(r"ns2vc::(\w+)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def family(k):
    if k.startswith(("gemm", "conv3ts", "ffn", "geglu", "rowchain", "splitk")):
        return "implicit_gemm"
    if k.startswith("attn"):
        return "attention"
    if k.startswith(("gn_", "ln_apply_op")):
        return "norm_stats"
    return "other"


def commit():
    try:
        return subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        return os.environ.get("NS2VC_COMMIT", "unknown (GPU box has no .git; see the commit that added this file)")


bench = None
try:                                   # r6: the unabridged record (bench.py --detail-json), one JSON document
    bench = json.load(open(f_bench))
except Exception:
    for line in open(f_bench):         # (r1-r5: the one long stdout line)
        line = line.strip()
        if line.startswith("{"):
            bench = json.loads(line)
fams = bench["roofline"]["families"]
per_step = {k: v["launches"] for k, v in fams.items()}
tot, calls = {}, {}
for r in csv.DictReader(open(f_stats)):
    k = short(r["Name"])
    if "ns2vc::" not in r["Name"]:
        continue
    f = family(k)
    tot[f] = tot.get(f, 0.0) + float(r["TotalDurationNs"])
    calls[f] = calls.get(f, 0) + int(r["Calls"])
# forward-equivalents in the trace: attn_kernel is launched only by the per-step plan (the condition pass pools with its own kernel);
# r4's last builds have no norm family left to count by
n_fwd = calls.get("attention", 0) / max(per_step.get("attention", 1), 1)
out = {"source": "rocprofv3 --kernel-trace --stats -- " + " ".join(["python bench.py"] + [f"--{k.replace('_', '-')} {v}" for k, v in (("steps", bench["steps"]), ("warmup", bench["warmup"]))])
                 + " --skip-cpu --reps 3",
       "precision": bench["dtype"], "shape": [bench["config"]["global_batch"], bench["config"]["frames"], bench["config"]["prompt_frames"]],
       "commit": commit(), "forward_equivalents_in_trace": n_fwd, "step_ms": bench["ms_per_step"],
       "families": {f: {"calls": calls[f], "ms_per_step": tot[f] / 1e6 / max(n_fwd, 1e-9)} for f in tot}}
out["kernel_ms_per_step_total"] = sum(v["ms_per_step"] for v in out["families"].values())
json.dump(out, open(os.path.join(outdir, f"{tag}_family_times.json"), "w"), indent=1)
print("wrote", tag, "family times:", {k: round(v["ms_per_step"], 3) for k, v in out["families"].items()}, "step", round(bench["ms_per_step"], 3))
