#!/bin/bash
# Builds the WHOLE library of a git ref side by side with the working tree's: gh-icp_amd/libghicp_var_<name>.so (git-ignored; travels to the GPU box
# with the snapshot), to be selected with GHICP_LIB=... for bench.py / tests / scripts.  Unlike scripts/km_variant_lib.sh (which recompiles the
# Kuhn-Munkres sources only) this works for refs that change any kernel or header.      usage: scripts/branch_lib.sh <git-ref> <name>
set -e
ref=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d)
git -C "$root" archive "$ref" gh-icp_amd/csrc include | tar -x -C "$d"
make -C "$d/gh-icp_amd/csrc" -j8 >/dev/null
cp "$d/gh-icp_amd/libghicp_hip.so" "$root/gh-icp_amd/libghicp_var_$name.so"
rm -rf "$d"
echo "built gh-icp_amd/libghicp_var_$name.so from $ref"
