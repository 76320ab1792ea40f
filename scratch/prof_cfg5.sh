#!/bin/bash
# rocprofv3 kernel stats of the configs[4] inference call (B = 64, eval, fp16)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-prof_cfg5}
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o c5 --output-format csv -- python3 $R/scratch/bench_infer.py > $out/log.txt 2>&1
tail -1 $out/log.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*kernel_trace.csv")[0]
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg[k][0]+=1; agg[k][1]+=d
tot=sum(v[1] for v in agg.values())
print("total kernel ms", round(tot/1e3,1))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:24]:
    print(f"{v[1]/v[0]:8.1f} us/launch n={v[0]:5d} {100*v[1]/tot:5.1f}%  {k[:110]}")
PY
