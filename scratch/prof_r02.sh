#!/bin/bash
# Round-2 evidence for the 4a bench line (run on the GPU box through gpurun): kernel-trace stats of the default bench command
# (streams overlapped, as timed) and of the same step with the expert / backbone streams serialised (a launch's duration is then
# its own), then the HBM-side traffic of the dominant kernel from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace
# only, eager, streams serialised).  Everything lands in gpurun_out/$1; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-prof_r02}
rm -rf $out; mkdir -p $out/overlap $out/serial
rocprofv3 --kernel-trace --stats -d $out/overlap -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-extras > $out/overlap/bench.log 2>&1
tail -1 $out/overlap/bench.log | cut -c1-200
export AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0
rocprofv3 --kernel-trace --stats -d $out/serial -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-extras > $out/serial/bench.log 2>&1
tail -1 $out/serial/bench.log | cut -c1-200
export AUTOMOE_PREFETCH_EXPERTS=0 AUTOMOE_HIPGRAPH=0
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $out/pmc_log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out="$out"
for mode in ("overlap","serial"):
    f=glob.glob(out+"/%s/*kernel_trace.csv"%mode)[0]
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in rows:
        k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        agg[k][0]+=1; agg[k][1]+=d
    tot=sum(v[1] for v in agg.values())
    print("== %s: total kernel time %.1f ms, %d distinct kernels"%(mode,tot/1e3,len(agg)))
    with open(out+"/%s_top.txt"%mode,"w") as fo:
        for k,(n,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:32]:
            line="%8.1f us/launch  n=%5d  %5.1f%%  %s"%(t/n,n,100*t/tot,k[:110])
            fo.write(line+"\n")
            if mode=="serial": print(line)
# PMC: per-launch rows of the dominant kernel (forward launches: the grid of a dgrad launch is recognisable by its K, not here: all kept)
res={}; raw=[]
for f in sorted(glob.glob(out+"/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_ring16_k<256, 256" not in k: continue
        res.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
        raw.append({"counter":r["Counter_Name"],"value":float(r["Counter_Value"]),"grid":r.get("Grid_Size",""),"dispatch":r.get("Dispatch_Id","")})
summary={c:{"launches":len(v),"mean":sum(v)/len(v),"min":min(v),"max":max(v)} for c,v in res.items()}
print(json.dumps(summary))
json.dump({"summary":summary,"rows":raw},open(out+"/pmc_ring16.json","w"))
PY
