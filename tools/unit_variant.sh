#!/bin/bash
# Build variants/libvbmc_<name>.so: the shipped library with ONE translation unit (SRC) recompiled under extra -D flags.
#   usage: SRC=gp_fused.hip tools/unit_variant.sh name "-DFLAG ..." [name flags ...]
set -e
cd "$(dirname "$0")/.."
python -m pyvbmc_amd.build > /dev/null
mkdir -p variants
OBJ=pyvbmc_amd/csrc/_obj
STEM=${SRC%.hip}
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags \
    -c pyvbmc_amd/csrc/$SRC -o variants/${STEM}_$name.o
  objs=$(ls $OBJ/*.o | grep -v "/$STEM.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/${STEM}_$name.o -o variants/libvbmc_$name.so -ldl -Wl,-rpath,/opt/rocm/lib
  rm variants/${STEM}_$name.o
  echo "variants/libvbmc_$name.so"
done
