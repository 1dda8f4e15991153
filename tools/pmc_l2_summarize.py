#!/usr/bin/env python3
"""Condense the rocprofv3 counter CSVs of tools/pmc_l2.sh into profiles/<tag>_pmc_l2.json: per kernel of the bench step, the L2 (TCC) hit
rate, the L2 request mix and the memory-side (EA = beyond L2: Infinity Cache / HBM) reads and writes per launch.
usage: pmc_l2_summarize.py TAG OUTDIR pass1.csv [pass2.csv ...]"""
import csv, json, os, re, subprocess, sys
from collections import defaultdict

tag, outdir, files = sys.argv[1], sys.argv[2], sys.argv[3:]


def short(name):
    m = re.search(r"ns2vc::(\w+)(<[^(]*>)?\(", name)
    if not m:
        return name.split("(")[0][:60]
    return m.group(1) + (m.group(2) or "").replace("ns2vc::", "")


def family(k):
    if k.startswith(("gemm", "conv3ts", "ffn", "geglu", "rowchain", "splitk", "tokchain")):
        return "implicit_gemm"
    if k.startswith("attn"):
        return "attention"
    return "other"


per = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> per-dispatch values
dur = defaultdict(dict)                            # kernel -> dispatch id -> us (first pass that has it)
for path in files:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if "at::" in k or k.startswith("__amd"):
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].setdefault((path, r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def commit():
    try:
        return subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short=12", "HEAD"], text=True,
                                       stderr=subprocess.DEVNULL).strip()
    except Exception:
        return os.environ.get("NS2VC_COMMIT", "unknown (GPU box has no .git; see the commit that added this file)")


def mean(c, name):
    v = c.get(name)
    return sum(v) / len(v) if v else None


kern, fam = {}, defaultdict(lambda: defaultdict(float))
for k, c in sorted(per.items()):
    n = max(len(v) for v in c.values())
    us = sum(dur[k].values()) / max(len(dur[k]), 1)
    hit, miss, req, rd, wr = (mean(c, x) for x in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum"))
    ea_rd, ea_32, ea_dram = mean(c, "TCC_EA0_RDREQ_sum"), mean(c, "TCC_EA0_RDREQ_32B_sum"), mean(c, "TCC_EA0_RDREQ_DRAM_sum")
    ea_wr, ea_wr64 = mean(c, "TCC_EA0_WRREQ_sum"), mean(c, "TCC_EA0_WRREQ_64B_sum")
    e = {"launches_counted": n, "avg_us_under_pmc": round(us, 2)}
    if hit is not None and miss is not None and hit + miss > 0:
        e["l2_hit_rate"] = round(hit / (hit + miss), 4)
        e["l2_hit_per_launch"], e["l2_miss_per_launch"] = round(hit), round(miss)
    if req is not None:
        e["l2_req_per_launch"] = round(req)
        e["l2_read_req_per_launch"], e["l2_write_req_per_launch"] = (round(rd) if rd is not None else None), (round(wr) if wr is not None else None)
    if ea_rd is not None:
        # memory-side reads: requests are 64 B unless flagged 32 B; gfx950 tallies a 128-B request as one (guide: FETCH_SIZE x2) -> both figures given
        b64 = (ea_rd - (ea_32 or 0.0)) * 64 + (ea_32 or 0.0) * 32
        e["ea_read_req_per_launch"], e["ea_read_req_32B_per_launch"] = round(ea_rd), round(ea_32 or 0)
        e["ea_read_mb_per_launch_at_64B"], e["ea_read_mb_per_launch_x2_corrected"] = round(b64 / 1e6, 3), round(2 * b64 / 1e6, 3)
        if ea_dram is not None:
            e["ea_read_req_dram_share"] = round(ea_dram / ea_rd, 4) if ea_rd else None
        if req:
            e["ea_reads_per_l2_read_req"] = round(ea_rd / rd, 4) if rd else None
    if ea_wr is not None:
        e["ea_write_req_per_launch"], e["ea_write_req_64B_per_launch"] = round(ea_wr), round(ea_wr64 or 0)
        e["ea_write_mb_per_launch"] = round(((ea_wr - (ea_wr64 or 0)) * 32 + (ea_wr64 or 0) * 64) / 1e6, 3)
    for name in ("TCC_TAG_STALL_sum", "TCC_STREAMING_REQ_sum", "TCC_NORMAL_EVICT_sum", "TCC_NORMAL_WRITEBACK_sum", "TCC_BUSY_sum", "TCC_CYCLE_sum",
                 "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_NC_READ_REQ_sum", "TCP_TCC_UC_READ_REQ_sum", "TCP_TCC_CC_READ_REQ_sum",
                 "TCP_TCC_RW_READ_REQ_sum"):
        v = mean(c, name)
        if v is not None:
            e[name.lower()[:-4] + "_per_launch"] = round(v)
    if e.get("tcc_busy_per_launch") and e.get("tcc_cycle_per_launch"):
        e["tcc_busy_share"] = round(e["tcc_busy_per_launch"] / e["tcc_cycle_per_launch"], 4)
    kern[k] = e
    f = fam[family(k)]
    for key, m in (("hit", hit), ("miss", miss), ("ea_rd", ea_rd), ("req", req)):
        if m is not None:
            f[key] += m * n
    f["n"] += n
families = {k: {"launches_counted": int(v["n"]), "l2_hit_rate": round(v["hit"] / (v["hit"] + v["miss"]), 4) if v["hit"] + v["miss"] > 0 else None,
                "l2_req_per_launch": round(v["req"] / v["n"]) if v["n"] else None} for k, v in fam.items()}
out = {"source": "rocprofv3 --pmc <4 TCC / TCP counters per pass> --kernel-trace -- python bench.py --skip-cpu --steps 4 --warmup 4 --reps 1 (tools/pmc_l2.sh); "
                 "hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2 section); EA = the L2's memory-side interface "
                 "(Infinity Cache + HBM behind it)", "workload": "10 s x batch 32, fp16, unipc", "precision": "fp16", "shape": [32, 938, 469],
       "commit": commit(), "families": families, "kernels": kern}
json.dump(out, open(os.path.join(outdir, f"{tag}_pmc_l2.json"), "w"), indent=1)
print(f"{'kernel':58s} {'us':>7s} {'L2 hit':>7s} {'L2 req/launch':>14s} {'EA rd MB(x2)':>12s} {'EA wr MB':>9s}")
for k, e in sorted(kern.items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["launches_counted"]):
    print(f"{k[:58]:58s} {e['avg_us_under_pmc']:7.1f} {e.get('l2_hit_rate', float('nan')):7.3f} {e.get('l2_req_per_launch', 0):14d} "
          f"{e.get('ea_read_mb_per_launch_x2_corrected', float('nan')):12.2f} {e.get('ea_write_mb_per_launch', float('nan')):9.2f}")
