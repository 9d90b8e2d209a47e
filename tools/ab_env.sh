#!/bin/bash
# A/B of bench.py under different ENVIRONMENT settings in one GPU session: tools/ab_env.sh OUTDIR "ENV1=.. ENV2=.." "..." (each variant twice, interleaved;
# a variant may also carry bench flags after a '|': "DVLP_P8_MIN_TILES=64|--overlap-wgrad 2")
out=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  j=0
  for v in "$@"; do
    envs="${v%%|*}"; flags=""; [[ "$v" == *"|"* ]] && flags="${v#*|}"
    env $envs python bench.py --no-cpu-baseline --no-kernel-timing $flags 2>"$out/err_${j}_$rep.txt" | tail -1 > "$out/bench_${j}_$rep.json"
    python - "$out/bench_${j}_$rep.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print("%-60s %7.3f ms/step  %7.1f pairs/s" % (sys.argv[2] or "(default)", d["ms_per_step"], d["value"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    j=$((j+1))
  done
done
