#!/bin/bash
# Round 5, BEFORE the first GPU call (runs in the build container, no GPU): the working tree's library plus one variant library per staged branch,
# side by side under gh-icp_amd/ (git-ignored, they travel to the GPU box with the snapshot).  ~4 minutes of hipcc.
#   occ5    next/pair-loop-96vgpr       k_pair_loop at 96 VGPRs (co-residency probe)
#   beside  next/fe-beside-slots        96-VGPR loop + 2 KB LDS buckets + small-LDS BSC / NMS (front end beside the slots)
#   besides next/fe-small-lds-primitives   the same + sort / select / unique in <= 6.2 KB of LDS when GHICP_FE_SMALL_LDS=1
#   packed  next/fe-packed-voxel-sort   keys-only voxel sort of the batched front end
#   sfused  next/km-s-rounds-fused      S rounds of the Kuhn-Munkres solver with one pass / one barrier per round
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C "$root/gh-icp_amd/csrc" -j8 >/dev/null
bash "$root/scripts/km_variant_lib.sh" next/pair-loop-96vgpr occ5
bash "$root/scripts/km_variant_lib.sh" next/km-s-rounds-fused sfused
bash "$root/scripts/branch_lib.sh" next/fe-beside-slots beside
bash "$root/scripts/branch_lib.sh" next/fe-small-lds-primitives besides
bash "$root/scripts/branch_lib.sh" next/fe-packed-voxel-sort packed
# the whole tree of the small-primitives branch with its library in place: its bench.py has --fe-small-lds {0,1,auto} (variants/ is git-ignored, it travels)
rm -rf "$root/variants/besides"; mkdir -p "$root/variants/besides"
git -C "$root" archive next/fe-small-lds-primitives | tar -x -C "$root/variants/besides"
cp "$root/gh-icp_amd/libghicp_var_besides.so" "$root/variants/besides/gh-icp_amd/libghicp_hip.so"
ls -l "$root"/gh-icp_amd/libghicp_*.so "$root"/variants/besides/gh-icp_amd/libghicp_hip.so
