#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q -s -k "gather_negatives_two_ranks" > gpurun_out/r6_t5a.txt 2>&1
grep -n "File\|line\|Thread\|Error\|passed\|failed" gpurun_out/r6_t5a.txt | tail -60 | cut -c1-300
python tools/loss_gemm_bench.py > gpurun_out/r6_loss_gemm2.txt 2>&1
XLOSS_ONLY=2 python tools/xloss_bench.py > gpurun_out/r6_xloss2.txt 2>&1
cat gpurun_out/r6_loss_gemm2.txt gpurun_out/r6_xloss2.txt
for i in 1 2; do
  for k in "" "--knob dvlp_dev_gemm_resident_b=0" "--knob dvlp_dev_gemm_p8_short_tiles=3" "--knob dvlp_dev_gemm_p8_short_tiles=3 --knob dvlp_dev_gemm_resident_b=0"; do
    python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"$k |\", d[\"ms_per_step\"], d[\"object_transformer\"][\"ms\"])"
  done
done > gpurun_out/r6_ab2.txt 2>&1
cat gpurun_out/r6_ab2.txt
