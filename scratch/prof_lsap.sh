#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_lsap
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o l --output-format csv -- python3 $R/scratch/ab_lsap.py > $out/log.txt 2>&1
tail -5 $out/log.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*kernel_trace.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "lsap" in k or "fill" in k.lower() or "memset" in k.lower():
        agg[(k[:70], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v); print("%8.1f us median  %8.1f max  n=%4d  %s grid %s,%s"%(v2[len(v2)//2], v2[-1], len(v), k[0], k[1], k[2]))
PY
