#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for e in 0 1; do
    DVLP_CORUN_WGRAD_SPLIT=$e python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"corun_wgrad_split=$e |\", d[\"ms_per_step\"], d[\"object_transformer\"][\"ms\"], d[\"config\"][\"final_loss\"])"
  done
done > gpurun_out/r6_ab5.txt 2>&1
cat gpurun_out/r6_ab5.txt
