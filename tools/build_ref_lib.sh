#!/bin/bash
# Build the library as of a git revision into demovlp_amd/lib/libdemovlp_hip_<name>.so (untracked; travels with gpurun) for in-session A/B
# through DEMOVLP_HIP_LIB:   tools/build_ref_lib.sh <rev> <name>
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" demovlp_amd/csrc | tar -x -C "$tmp"
objs=""
for f in "$tmp"/demovlp_amd/csrc/*.hip; do
  o="$tmp/$(basename "${f%.hip}").o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/demovlp_amd/lib/libdemovlp_hip_$name.so" $objs
rm -rf "$tmp"
echo "$root/demovlp_amd/lib/libdemovlp_hip_$name.so"
