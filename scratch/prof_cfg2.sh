#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-prof_cfg2}
rm -rf $out; mkdir -p $out
cat > /tmp/run_cfg2.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, bench
from self_driving_model_amd import runtime
runtime.set_compute_dtype(torch.float16)
print("cfg2 img/s", bench.bench_drivable(16, 6, 3))
PY
rocprofv3 --kernel-trace --stats -d $out -o c2 --output-format csv -- python3 /tmp/run_cfg2.py > $out/log.txt 2>&1
tail -2 $out/log.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*kernel_trace.csv")[0]
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg[k][0]+=1; agg[k][1]+=d
tot=sum(v[1] for v in agg.values())
print("total kernel ms %.1f"%(tot/1e3))
for k,(n,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:22]:
    print("%8.1f us/launch n=%5d %5.1f%%  %s"%(t/n,n,100*t/tot,k[:100]))
PY
