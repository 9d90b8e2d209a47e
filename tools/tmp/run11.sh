#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "riding or wgrad_grouped or graph_replay_equals_eager or bit_reproducible or adamw or fused_adamw" > gpurun_out/r6_t11.txt 2>&1
tail -6 gpurun_out/r6_t11.txt | cut -c1-300
for i in 1 2 3; do
  for e in 1 0; do
    python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --knob dvlp_dev_wgrad_group_ride=$e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"ride=$e |\", d[\"ms_per_step\"], d[\"object_transformer\"][\"ms\"], d[\"config\"][\"final_loss\"])"
  done
done > gpurun_out/r6_ab6.txt 2>&1
cat gpurun_out/r6_ab6.txt
