#!/bin/bash
# PMC passes for the layer1 conv kernel (one counter group per run; kernel-trace only, as the pool requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_${SHAPE:-l1}
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p$i --output-format csv -- python3 $R/scratch/one_conv.py ${SHAPE:-l1} > $out/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$out/p*/*counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    agg=collections.OrderedDict()
    for r in rows:
        k=r["Kernel_Name"][:60]
        if "conv3x3" not in k and "conv_" not in k: continue
        agg.setdefault((k,r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for (k,c),v in agg.items():
        print(k,c,"last=%.4g"%v[-1],"n=%d"%len(v))
PY
