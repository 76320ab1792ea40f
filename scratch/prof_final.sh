#!/bin/bash
# Round-end evidence: kernel-trace stats of the default bench (streams overlapped, as timed) and of the same step with the
# expert / backbone streams serialised (per-kernel durations without co-running kernels); summaries go to gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-prof_final2}
rm -rf $out; mkdir -p $out/overlap $out/serial
rocprofv3 --kernel-trace --stats -d $out/overlap -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-extras > $out/overlap/bench.log 2>&1
tail -1 $out/overlap/bench.log | cut -c1-200
export AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0
rocprofv3 --kernel-trace --stats -d $out/serial -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-extras > $out/serial/bench.log 2>&1
tail -1 $out/serial/bench.log | cut -c1-200
python3 - <<PY
import csv,glob,collections
for mode in ("overlap","serial"):
    f=glob.glob("$out/%s/*kernel_trace.csv"%mode)[0]
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda:[0,0.0])
    t0=min(int(r["Start_Timestamp"]) for r in rows); t1=max(int(r["End_Timestamp"]) for r in rows)
    for r in rows:
        k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        agg[k][0]+=1; agg[k][1]+=d
    tot=sum(v[1] for v in agg.values())
    print("== %s: total kernel time %.1f ms, %d distinct kernels"%(mode,tot/1e3,len(agg)))
    with open("$out/%s_top.txt"%mode,"w") as fo:
        for k,(n,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:30]:
            line="%8.1f us/launch  n=%5d  %5.1f%%  %s"%(t/n,n,100*t/tot,k[:100])
            fo.write(line+"\n")
            if mode=="serial": print(line)
PY
