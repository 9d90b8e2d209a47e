#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/loss_gemm_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r6_loss_gemm3.txt; cut -c1-150 gpurun_out/r6_loss_gemm3.txt
XLOSS_ONLY=2 python tools/xloss_bench.py 2>&1 | grep -v amdgpu
python -m pytest tests -m gpu -x -q -k "xattn or local_loss or gemm or loss_heads" > gpurun_out/r6_t15.txt 2>&1; tail -3 gpurun_out/r6_t15.txt | cut -c1-200
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"new |\", d[\"ms_per_step\"], d[\"object_transformer\"][\"ms\"])"; done
