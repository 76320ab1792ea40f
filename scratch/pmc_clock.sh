#!/bin/bash
# Average shader clock every kernel of the (serialised, eager) 4a step holds: GRBM_GUI_ACTIVE cycles per dispatch / its duration.
# Output: gpurun_out/pmc_clock/clocks.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_clock
rm -rf $out; mkdir -p $out
export AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0 AUTOMOE_PREFETCH_EXPERTS=0 AUTOMOE_HIPGRAPH=0
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $out/p -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $out/log.txt 2>&1
python3 - <<PY > $out/clocks.txt
import csv,glob,collections
tr={}
for f in glob.glob("$out/p/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): tr[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for f in glob.glob("$out/p/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
        d=tr.get(r["Dispatch_Id"])
        if not d: continue
        a=agg[r["Kernel_Name"][:70]]; a[0]+=1; a[1]+=float(r["Counter_Value"]); a[2]+=d
for k,(n,c,d) in sorted(agg.items(), key=lambda kv:-kv[1][2])[:30]:
    print(f"{d/n/1e3:8.1f} us/launch n={n:4d}  cycles/ns = {c/d:6.3f}  {k}")
PY
cat $out/clocks.txt
