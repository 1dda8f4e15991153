"""summarise a rocprofv3 counter_collection.csv of tools/pmc_lds.sh: per kernel, LDS bank-conflict cycles / LDS-active cycles"""
import collections, csv, json, os, re, sys

src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for r in csv.DictReader(open(src)):
    k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("ns2vc::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        launches[k] += 1
out = {"source": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -- python bench.py --skip-cpu "
                 "--skip-fp32 --steps 4 --warmup 4 --reps 1 (own pass)", "commit": os.environ.get("NS2VC_COMMIT"), "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    a, c = v.get("SQ_ACTIVE_INST_LDS", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    if not k.startswith(("gemm", "conv3ts", "attn", "ffn", "geglu", "rowchain", "gn_apply", "time_embed", "solver")):
        continue
    out["kernels"][k] = {"launches": launches[k], "lds_insts": v.get("SQ_INSTS_LDS", 0.0), "lds_active_cycles": a, "bank_conflict_cycles": c,
                         "conflict_fraction_of_lds_active": round(c / a, 4) if a else None}
    print(f"{k[:70]:70s} conflict / LDS-active = {c / a if a else 0:.3f}")
json.dump(out, open(dst, "w"), indent=1)
