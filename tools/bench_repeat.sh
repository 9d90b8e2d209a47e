#!/bin/bash
# Run-to-run spread of the default bench line on ONE box: N back-to-back `python bench.py` processes (default 3).
# Usage (on the GPU box): bash tools/bench_repeat.sh [N] > gpurun_out/bench_repeat.txt
cd "$(dirname "$0")/.."
N=${1:-3}
for i in $(seq 1 "$N"); do
    python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('run $i: %.3f ms/step  %.1f pairs/s  object tower %.2f ms  text tower %.2f ms  loss heads %.2f ms' % (
    d['ms_per_step'], d['value'], d['roofline']['object_transformer_ms'], d['roofline']['text_tower_ms'], d['roofline']['loss_heads_ms']))"
done
