#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"early-embed |\", d[\"ms_per_step\"], d[\"object_transformer\"][\"ms\"], d[\"config\"][\"final_loss\"])"
  DVLP_NO_EARLY_EMBED_UPDATE=1 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"tail-embed  |\", d[\"ms_per_step\"], d[\"object_transformer\"][\"ms\"], d[\"config\"][\"final_loss\"])"
done > gpurun_out/r6_ab7.txt 2>&1
cat gpurun_out/r6_ab7.txt
python -m pytest tests -m gpu -x -q -k "graph_replay_equals_eager or bit_reproducible or adamw or riding or space_attention or ten_step" > gpurun_out/r6_t13.txt 2>&1
tail -4 gpurun_out/r6_t13.txt | cut -c1-200
