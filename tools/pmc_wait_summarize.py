"""summarise a rocprofv3 counter_collection.csv of tools/pmc_wait.sh: per kernel, where the waves' cycles go (SQ_WAVE_CYCLES split into
parked at s_waitcnt / barrier = SQ_WAIT_ANY, issue stalls = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY; MI355X_MICROARCH.md PMC slots) and the
VALU / transcendental / MFMA instruction mix.  r5."""
import collections, csv, json, os, re, sys

src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
names = set()
for r in csv.DictReader(open(src)):
    k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("ns2vc::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names.add(r["Counter_Name"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        launches[k] += 1
out = {"source": "rocprofv3 --pmc " + " ".join(sorted(names)) + " --kernel-trace -- python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 4 --warmup 4 --reps 1 (own pass)",
       "commit": os.environ.get("NS2VC_COMMIT"), "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if not k.startswith(("gemm", "conv3ts", "attn", "ffn", "geglu", "rowchain")):
        continue
    wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    row = {"launches": launches[k]}
    for c in sorted(names):
        row[c] = v.get(c, 0.0)
    row["parked_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4)
    row["issue_stall_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
    row["issuing_fraction (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
    row["valu_active_fraction (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)"] = round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
    if v.get("SQ_INSTS_VALU"):
        row["trans_share_of_valu_insts"] = round(v.get("SQ_INSTS_VALU_TRANS", 0.0) / v["SQ_INSTS_VALU"], 4)
    out["kernels"][k] = row
    print(f"{k[:64]:64s} parked {row['parked_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)']:.3f}  issue-stall {row['issue_stall_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)']:.3f}  "
          f"issuing {row['issuing_fraction (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)']:.3f}  VALU {row['valu_active_fraction (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)']:.3f}")
json.dump(out, open(dst, "w"), indent=1)
