#!/bin/bash
# A/B of bench.py over several builds of the library in one GPU session (each build twice, interleaved):
#   tools/ab_libs_bench.sh OUTFILE lib1.so lib2.so ...      (paths relative to demovlp_amd/lib/; tools/build_ref_lib.sh makes reference builds)
out=$1; shift
: > "$out"
for rep in 1 2; do
  for L in "$@"; do
    DEMOVLP_HIP_LIB=$PWD/demovlp_amd/lib/$L python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline', {}); o=d.get('object_transformer', {})
print('%-28s %7.3f ms/step %7.1f pairs/s   object tower %7.3f ms  frac %.4f' % ('$L', d['ms_per_step'], d['value'], r.get('object_transformer_ms') or o.get('ms') or 0, r.get('object_transformer_frac') or o.get('frac') or 0))" | tee -a "$out"
  done
done
