#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r6_suite3.txt 2>&1
tail -12 gpurun_out/r6_suite3.txt | cut -c1-300
