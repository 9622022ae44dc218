EBCC_LOG_LEVEL=0 EBCC_HIP_SLICES=1 python bench.py --frames 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2> gpurun_out/trace.txt > /dev/null
python - <<'PY'
import re,collections
per=collections.defaultdict(lambda: [0,0]); crs=collections.defaultdict(list)
for l in open('gpurun_out/trace.txt', errors='ignore'):
    m=re.search(r"frame (\d+) \(search (\d)\): cr ([0-9.]+) 1-quantile ([0-9.e+-]+)", l)
    if m:
        f,k,cr=int(m.group(1)),int(m.group(2)),float(m.group(3)); per[f][k]+=1; crs[(f,k)].append((cr,float(m.group(4))))
h0=collections.Counter(v[0] for v in per.values()); h1=collections.Counter(v[1] for v in per.values())
print("search0 real probes per frame:", sorted(h0.items())); print("search1 real probes per frame:", sorted(h1.items()))
print("frame 0 search 0:", crs[(0,0)]); print("frame 0 search 1:", crs[(0,1)])
PY
