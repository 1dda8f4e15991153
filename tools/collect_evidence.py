#!/usr/bin/env python3
"""Copy the evidence set `tools/sessions/r6_final.sh` left under gpurun_out/ into profiles/ (the tracked, judged copies), stamping the commit it was measured at:
    python tools/collect_evidence.py <commit>
Writes profiles/r06_bench_driver_cmd.json (the driver command's stdout line + its bench_detail.json), r06_bench_full.json, the rocprofv3 / PMC summaries, the L2 table and
the determinism probe's output."""
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
P = os.path.join(R, "profiles")


def lastline(p):
    return json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])


def wall(p):
    m = re.search(r"real\s+(\d+)m([\d.]+)s", open(p).read())
    return int(m.group(1)) * 60 + float(m.group(2))


def main():
    commit = sys.argv[1]
    f = os.path.join(G, "final")
    l1, l2 = lastline(f + "/bench_driver_cmd_1.json"), lastline(f + "/bench_driver_cmd_2.json")
    w1, w2 = wall(f + "/bench_driver_cmd_1.err"), wall(f + "/bench_driver_cmd_2.err")
    json.dump({"command": f"python bench.py --gpus 1 --steps 20 --warmup 5   (the driver's command; {w1:.1f} s wall on the GPU box; commit {commit})",
               "stdout_line": l1, "stdout_line_bytes": len(json.dumps(l1)),
               "second_run_same_box": {"value": l2["value"], "ms_per_step": l2["ms_per_step"], "wall_s": w2},
               "bench_detail_json": json.load(open(f + "/bench_detail_driver_cmd.json"))}, open(P + "/r06_bench_driver_cmd.json", "w"), indent=1)
    lf, wf = lastline(f + "/bench_full_line.json"), wall(f + "/bench_full.err")
    json.dump({"command": f"python bench.py --full   ({wf:.0f} s wall; the compact line is printed before the long legs and again as the last line; commit {commit})",
               "bench_detail_json": json.load(open(f + "/bench_full_detail.json")), "stdout_last_line": lf}, open(P + "/r06_bench_full.json", "w"), indent=1)
    for n in ("r06_family_times.json", "r06_kernel_stats_bench_steps20.csv", "r06_pmc_hbm_traffic.json", "r06_pmc_l2.json", "r06_pmc_mfma_util.json", "r06_stats_bench.json"):
        shutil.copy(os.path.join(G, n), os.path.join(P, n))
    shutil.copy(f + "/pmc_l2.txt", P + "/r06_pmc_l2_table.txt")
    shutil.copy(f + "/determinism.txt", P + "/r06_determinism_probe_25loops.txt")
    for n in ("r06_family_times.json", "r06_pmc_hbm_traffic.json", "r06_pmc_l2.json", "r06_pmc_mfma_util.json"):
        p = os.path.join(P, n)
        d = json.load(open(p))
        if "commit" in d:
            d["commit"] = f"{commit} (measured on the GPU box, which has no .git)"
        json.dump(d, open(p, "w"), indent=1)
    ft = json.load(open(P + "/r06_family_times.json"))
    print(f"{commit}: {l1['ms_per_step']:.3f} ms/step, value {l1['value']:.1f} ({w1:.1f} s wall, {len(json.dumps(l1))} B); second run {l2['ms_per_step']:.3f}; "
          f"rocprof step {ft['step_ms']:.3f} ms, gemm family {ft['families']['implicit_gemm']['ms_per_step']:.3f} ms; "
          f"parity {l1['parity']['rel_l2_vs_oracle']:.3e} / {l1['parity']['sampled_latent']:.3e}; roofline frac {l1['roofline']['frac']}")


if __name__ == "__main__":
    main()
