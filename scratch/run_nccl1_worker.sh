#!/bin/bash
# Runs the one-rank RCCL worker of tests/test_hip_multigpu.py directly and keeps its whole output (the test only shows the tail).
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<PY
import sys
sys.path.insert(0, "$R/tests"); sys.path.insert(0, "$R")
import importlib.util
spec = importlib.util.spec_from_file_location("m", "$R/tests/test_hip_multigpu.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
open("/tmp/nccl1_worker.py", "w").write(m._NCCL1_WORKER)
PY
MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29679 /tmp/nccl1_worker.py $R
