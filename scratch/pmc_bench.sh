#!/bin/bash
# HBM-side traffic of the dominant kernel inside the real bench step: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (kernel-trace only), streams serialised so each dispatch's counters are its own.  Output: gpurun_out/pmc_bench/summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_bench
rm -rf $out; mkdir -p $out
export AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0 AUTOMOE_PREFETCH_EXPERTS=0 AUTOMOE_HIPGRAPH=0
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $out/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections,json
res={}
for f in sorted(glob.glob("$out/p*/*counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    for r in rows:
        k=r["Kernel_Name"]
        if "conv_ring_k<256, 256" not in k: continue
        d=res.setdefault(r["Counter_Name"],[])
        d.append(float(r["Counter_Value"]))
summary={c:{"launches":len(v),"mean":sum(v)/len(v),"min":min(v),"max":max(v)} for c,v in res.items()}
print(json.dumps(summary))
json.dump(summary,open("$out/summary.json","w"))
PY
