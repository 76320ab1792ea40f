#!/bin/bash
# Round-3 evidence for the 4a bench line (run on the GPU box through gpurun; the program always directly after `--`):
#   1. kernel-trace stats of the default bench command (streams overlapped, as timed) and of the same step with the expert /
#      backbone streams serialised (a launch's duration is then its own);
#   2. HBM-side traffic of the step's conv kernels: FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (kernel-trace only,
#      eager, streams serialised);
#   3. issue-side counters (MFMA busy, LDS wait, VMEM issue) in further separate passes.
# Everything lands in gpurun_out/$1; `python3 scratch/prof_r03_summarise.py gpurun_out/$1 profiles/r03` turns it into the tracked files.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-prof_r03}
rm -rf $out; mkdir -p $out/overlap $out/serial
rocprofv3 --kernel-trace --stats -d $out/overlap -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-extras > $out/overlap/bench.log 2>&1
tail -1 $out/overlap/bench.log | cut -c1-200
export AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0
rocprofv3 --kernel-trace --stats -d $out/serial -o bench --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-extras > $out/serial/bench.log 2>&1
tail -1 $out/serial/bench.log | cut -c1-200
export AUTOMOE_PREFETCH_EXPERTS=0 AUTOMOE_HIPGRAPH=0
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $out/pmc_log$i.txt 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc_log$i.txt; }
  echo "pmc pass $i done"
done
python3 $R/scratch/prof_r03_summarise.py $out $out/summary
ls $out/summary
