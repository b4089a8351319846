#!/bin/bash
# Registers, scratch and LDS of every kernel of one object file of the build (or of the whole library):
#   tools/kernel_regs.sh pyvbmc_amd/csrc/_obj/entropy_ws_dp10.o [name filter]
set -e
OBJ=${1:-pyvbmc_amd/libvbmc_hip.so}
FILTER=${2:-.}
B=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$B/llvm-objcopy -O binary --only-section=.hip_fatbin "$OBJ" $tmp/fat.bin
tgt=$($B/clang-offload-bundler --list --type=o --input=$tmp/fat.bin | grep gfx950 | head -1)
$B/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=$tgt --output=$tmp/k.co
$B/llvm-readelf --notes $tmp/k.co | awk '
  /\.agpr_count:/ {ag=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {sc=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2}
  /\.group_segment_fixed_size:/ {lds=$2} /\.wavefront_size:/ {print "vgpr", vg, "agpr", ag, "sgpr", sg, "scratch", sc, "lds", lds, name}' | c++filt | grep -E "$FILTER"
rm -rf $tmp
