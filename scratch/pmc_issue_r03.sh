#!/bin/bash
# Round 3: issue-side PMC passes (MFMA busy, LDS wait, VMEM issue) for the step's three biggest conv kernels INSIDE the real
# 4a bench step (eager, streams serialised so a dispatch's counters are its own).  One counter group per run, kernel-trace only,
# program directly after `--`.  Output: gpurun_out/$1/issue_counters.json + .md (copy into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-pmc_issue_r03}
rm -rf $out; mkdir -p $out
export AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0 AUTOMOE_PREFETCH_EXPERTS=0 AUTOMOE_HIPGRAPH=0
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $out/log$i.txt 2>&1 || { echo "pass $i failed"; tail -5 $out/log$i.txt; }
  echo "pass $i done"
done
python3 - <<PY
import csv,glob,collections,json
out="$out"
want=("conv_ring16_k","conv3x3_c64n64_duo","conv_halo_k","conv_ring_k<256, 128","conv_s2d_pool_k","conv_s2d_k")
agg=collections.OrderedDict()
for f in sorted(glob.glob(out+"/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        tag=next((w for w in want if w in k),None)
        if tag is None: continue
        agg.setdefault(tag,collections.OrderedDict()).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
res={}
for tag,cs in agg.items():
    res[tag]={c:{"launches":len(v),"mean":sum(v)/len(v)} for c,v in cs.items()}
json.dump(res,open(out+"/issue_counters.json","w"),indent=1)
with open(out+"/issue_counters.md","w") as fo:
    for tag,cs in res.items():
        fo.write("### %s\n\n| counter | launches | mean per launch |\n|---|---|---|\n"%tag)
        for c,d in cs.items(): fo.write("| %s | %d | %.4g |\n"%(c,d["launches"],d["mean"]))
        g=lambda n: cs.get(n,{}).get("mean")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
            fo.write("\nMFMA busy / SQ busy = %.3f\n"%(g("SQ_VALU_MFMA_BUSY_CYCLES")/g("SQ_BUSY_CYCLES")))
        fo.write("\n")
print(open(out+"/issue_counters.md").read()[:3000])
PY
