#!/bin/bash
# Build variants/libvbmc_<name>.so: the shipped library with the wave-split entropy kernel of ONE padded D
# (DP, default 10) recompiled under extra -D flags (ablations, in-kernel timestamps, scheduling experiments).
#   usage: [DP=10] tools/ws_variant.sh name "-DFLAG ..." [name flags ...]
set -e
cd "$(dirname "$0")/.."
python -m pyvbmc_amd.build > /dev/null
mkdir -p variants
OBJ=pyvbmc_amd/csrc/_obj
DP=${DP:-10}
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DVBMC_DP=$DP $flags \
    -c pyvbmc_amd/csrc/entropy_ws.hip -o variants/entropy_ws_dp${DP}_$name.o
  objs=$(ls $OBJ/*.o | grep -v "/entropy_ws_dp$DP.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/entropy_ws_dp${DP}_$name.o -o variants/libvbmc_$name.so -ldl -Wl,-rpath,/opt/rocm/lib
  rm variants/entropy_ws_dp${DP}_$name.o
  echo "variants/libvbmc_$name.so"
done
