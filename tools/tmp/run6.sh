#!/bin/bash
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gather_rank*.log
DVLP_GATHER_DEBUG=$GRAFT_REPO_ROOT/gpurun_out timeout 400 python -m pytest tests -m gpu -x -q -s -k "gather_negatives_two_ranks" > gpurun_out/r6_t6a.txt 2>&1
tail -5 gpurun_out/r6_t6a.txt | cut -c1-200
for r in 0 1; do echo "== rank $r"; tail -12 gpurun_out/gather_rank$r.log; done
for i in 1 2; do
  for k in "" "--knob dvlp_dev_gemm_p8_short_tiles=3"; do
    python bench.py --no-cpu-baseline --steps 20 $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d[\"roofline\"]; print(\"$k |\", d[\"ms_per_step\"], r[\"object_transformer_ms\"], r[\"text_tower_ms\"], r[\"loss_heads_ms\"], r[\"frac\"])"
    python bench.py --no-cpu-baseline --steps 20 --parallel-towers 0 $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d[\"roofline\"]; print(\"one-stream $k |\", d[\"ms_per_step\"], r[\"object_transformer_ms\"], r[\"text_tower_ms\"], r[\"loss_heads_ms\"], r[\"frac\"])"
  done
done > gpurun_out/r6_ab3.txt 2>&1
cat gpurun_out/r6_ab3.txt
