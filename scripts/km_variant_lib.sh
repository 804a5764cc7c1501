#!/bin/bash
# Builds gh-icp_amd/libghicp_var_<name>.so from the Kuhn-Munkres sources (km4_dev.h and what includes it) of a git ref, linked with the
# other objects of the working tree's build -- a complete library that `python scripts/km_bench.py --lib <file> --more --check` can time, so
# that several variants of the solver cost ONE gpurun call (round 3: profiles/r03_km4_second_half.txt).  The .so files are git-ignored and
# travel to the GPU box with the snapshot.        usage: scripts/km_variant_lib.sh <git-ref> <name>
set -e
ref=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
make -C "$root/gh-icp_amd/csrc" -j8 >/dev/null
d=$(mktemp -d)
git -C "$root" archive "$ref" gh-icp_amd/csrc include | tar -x -C "$d"
mkdir -p "$d/gh-icp_amd/csrc/build"
for f in km km4 km_dense_door loop; do
  ( cd "$d/gh-icp_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -c $f.hip -o build/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/gh-icp_amd/libghicp_var_$name.so" \
  $(ls "$root"/gh-icp_amd/csrc/build/*.o | grep -v "/km.o\|/km4.o\|/km_dense_door.o\|/loop.o") "$d"/gh-icp_amd/csrc/build/{km,km4,km_dense_door,loop}.o
rm -rf "$d"
echo "built gh-icp_amd/libghicp_var_$name.so from $ref"
