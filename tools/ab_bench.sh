#!/bin/bash
# A/B of bench.py in one GPU session: tools/ab_bench.sh OUTDIR "flags A" "flags B" ...   (each variant runs twice, interleaved)
out=$1; shift
mkdir -p "$out"
i=0
for rep in 1 2; do
  j=0
  for v in "$@"; do
    python bench.py --no-cpu-baseline $v 2>"$out/err_${j}_$rep.txt" | tail -1 > "$out/bench_${j}_$rep.json"
    python - "$out/bench_${j}_$rep.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print("%-50s %7.3f ms/step  %7.1f pairs/s  gemm frac %.4f  object tower %s ms frac %s" % (sys.argv[2] or "(default)", d["ms_per_step"], d["value"], r.get("frac", 0), r.get("object_transformer_ms"), r.get("object_transformer_frac")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    j=$((j+1))
  done
done
