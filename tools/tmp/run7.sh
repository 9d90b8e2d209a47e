#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q -s -k "gather_negatives_two_ranks" > gpurun_out/r6_t7a.txt 2>&1
tail -5 gpurun_out/r6_t7a.txt | cut -c1-300
python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d[\"roofline\"]; print(\"default |\", d[\"ms_per_step\"], r[\"object_transformer_ms\"], r[\"text_tower_ms\"], r[\"loss_heads_ms\"], r[\"frac\"])"
python -m pytest tests -m gpu -x -q > gpurun_out/r6_suite2.txt 2>&1
tail -15 gpurun_out/r6_suite2.txt | cut -c1-300
