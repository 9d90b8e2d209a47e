#!/usr/bin/env python
"""Per-kernel register / spill / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
    python tools/kres.py demovlp_amd/csrc/gemm.hip [name-filter] [-DFLAG ...]"""
import re, subprocess, sys
src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *extra,
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
cur, rows = None, []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
for x, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    if flt and not any(f in n for f in flt):
        continue
    print(f"{n[:70]:70s} VGPR {x.get('VGPRs','?'):>4} AGPR {x.get('AGPRs','?'):>4} SGPR {x.get('TotalSGPRs','?'):>4} vspill {x.get('VGPRs Spill','?'):>4} sspill {x.get('SGPRs Spill','?'):>3} scratch {x.get('ScratchSize [bytes/lane]','?'):>5} occ {x.get('Occupancy [waves/SIMD]','?')}")
if r.returncode:
    print(r.stderr[-3000:])
