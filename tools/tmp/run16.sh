#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r6_suite5.txt 2>&1
tail -4 gpurun_out/r6_suite5.txt | cut -c1-200
bash tools/profile_round.sh r6_final c26f02ab0d5f > gpurun_out/r6_profile_round.log 2>&1
tail -3 gpurun_out/prof_round/bench.json | cut -c1-400
