mkdir -p gpurun_out/s18
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in head default; do
  if [ $v = head ]; then export NS2VC_LIB=$R/ns2vc_amd/lib/variants/head/libns2vc_hip.so; else unset NS2VC_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -- python $R/bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 > /dev/null 2>&1
  cp "$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/s18/stats_$v.csv
done
cd $R
python - <<'PY'
import csv
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['Name'][:90]]=(int(r['Calls']),float(r['TotalDurationNs']))
    return d
a,b=load('gpurun_out/s18/stats_head.csv'),load('gpurun_out/s18/stats_default.csv')
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0)); cb,tb=b.get(k,(0,0))
    rows.append((tb-ta,k,ca,ta,cb,tb))
rows.sort(key=lambda r:-abs(r[0]))
for r in rows[:14]:
    print(f"{r[0]/1e3:10.1f} us  {r[1][:70]:70s} calls {r[2]}/{r[4]}  avg {r[3]/max(r[2],1)/1e3:.2f} -> {r[5]/max(r[4],1)/1e3:.2f} us")
print('total', sum(v[1] for v in a.values())/1e6, sum(v[1] for v in b.values())/1e6)
PY
