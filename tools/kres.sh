#!/bin/bash
# kernel resource usage of one csrc file:  tools/kres.sh attention.hip [grep pattern]
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c /root/repo/demovlp_amd/csrc/$1 -o /tmp/kres_$$.o 2>&1 | python3 -c "
import sys, re
cur = None
rows = []
for l in sys.stdin:
    m = re.search(r'remark: (.*?) \[-Rpass', l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
for r in rows:
    if pat and not pat.search(r['name']): continue
    print('%-70s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %s sgpr-spill %4s vgpr-spill %4s' % (r['name'][:70], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('SGPRs Spill'), r.get('VGPRs Spill')))
" "${2:-.}"
rm -f /tmp/kres_$$.o
