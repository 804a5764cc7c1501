#!/bin/bash
# ISA of one kernel of one translation unit (development aid): scripts/isa.sh batch.hip k_fb_pca_cells [out.s]
# prints instruction count, VGPRs, scratch; writes the kernel's ISA to out.s (default /tmp/<kernel>.s)
src=$1; k=$2; out=${3:-/tmp/$k.s}
cd "$(dirname "$0")/../gh-icp_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o /tmp/_tu.s "$src" 2> /tmp/_tu.err || { grep -v warning /tmp/_tu.err | head -20; exit 1; }
S=$(grep -n "^_Z[A-Za-z0-9_]*$k[A-Za-z0-9_]*:" /tmp/_tu.s | head -1 | cut -d: -f1)
[ -z "$S" ] && { echo "kernel $k not found"; exit 1; }
E=$(awk -v s=$S 'NR>s && /s_endpgm/{print NR; exit}' /tmp/_tu.s)
sed -n "${S},${E}p" /tmp/_tu.s > "$out"
echo "instructions: $(grep -c '^\s*[vs]_\|^\s*ds_\|^\s*global_\|^\s*flat_\|^\s*buffer_\|^\s*scratch_' "$out")  flat_load: $(grep -c flat_load "$out")  scratch ops: $(grep -c 'scratch_' "$out")"
grep -n "\.num_vgpr\|\.private_seg_size" /tmp/_tu.s | grep "$k" | sed 's/.*\.\(num_vgpr\|private_seg_size\), /\1 /' | head -3
