#!/bin/bash
# rocprofv3 kernel trace of the default bench (config 4a); prints per-kernel time per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-prof}
rm -rf $out; mkdir -p $out
STEPS=5
rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python3 $R/bench.py --steps $STEPS --warmup 3 --no-extras > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-160
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
# keep only the timed region: last STEPS graph replays ~ approximate by taking total / (steps+warmup) is wrong; use all launches and divide by their count per step
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg[k][0]+=1; agg[k][1]+=d
nsteps=$STEPS+3+2   # warmup 3 + capture-related eager steps (approx)
tot=sum(v[1] for v in agg.values())
print("total kernel time %.1f ms over run; distinct kernels %d"%(tot/1e3,len(agg)))
for k,(n,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:28]:
    print("%7.1f us/launch  n=%5d  %5.1f%%  %s"%(t/n,n,100*t/tot,k[:90]))
PY
