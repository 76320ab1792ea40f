#!/bin/bash
# issue-side PMC passes for the ring kernel (SHAPE=l3|l4); one counter group per run, kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_ring_${SHAPE:-l4}
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p$i --output-format csv -- python3 $R/scratch/one_conv.py ${SHAPE:-l4} > $out/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$out/p*/*counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    agg=collections.OrderedDict()
    for r in rows:
        k=r["Kernel_Name"][:40]
        if "conv_ring" not in k: continue
        agg.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for (k,c),v in agg.items():
        print(c,"last=%.4g"%v[-1],"n=%d"%len(v))
PY
