#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -s -k "bf16_finetune_then" > gpurun_out/r6_t8a.txt 2>&1
grep -n "G14\|bf16:\|passed\|failed\|bad" gpurun_out/r6_t8a.txt | cut -c1-300
