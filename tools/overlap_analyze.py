#!/usr/bin/env python3
"""usage: overlap_analyze.py kernel_trace.csv run.json -- how much of the PyTorch stages' kernel time (front end, vocoder: every
kernel that is not ns2vc::*) ran while a denoiser kernel was also in flight, from a rocprofv3 --kernel-trace CSV of
tools/overlap_run.py.  Only the timed batches are analysed: the last (ms_per_batch x batches) of the trace, as run.json reports them
(the warm-up batch and the setup are in front)."""
import csv
import json
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "ns2vc::" in r["Kernel_Name"]))
rows.sort()
run = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
t_hi = max(r[1] for r in rows)
t_lo = t_hi - int(run["ms_per_batch"] * run["batches"] * 1e6 * 1.01)
rows = [r for r in rows if r[0] >= t_lo]
den = [(s, e) for s, e, d in rows if d]
oth = [(s, e) for s, e, d in rows if not d]


def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            out += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


ud, uo = union(den), union(oth)
span = rows[-1][1] - rows[0][0]
both = intersect(ud, uo)
print(json.dumps({"span_ms": span / 1e6, "denoiser_kernel_busy_ms": total(ud) / 1e6, "pytorch_kernel_busy_ms": total(uo) / 1e6,
                  "both_in_flight_ms": both / 1e6, "fraction_of_pytorch_kernel_time_overlapped": both / max(total(uo), 1),
                  "device_idle_ms": (span - total(union(den + oth))) / 1e6, "denoiser_kernels": len(den), "pytorch_kernels": len(oth)}))
