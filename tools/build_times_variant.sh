#!/bin/bash
# variants/libvbmc_st.so: the shipped library with prep.hip, entropy.hip and api_elbo.hip rebuilt with
# -DFIN_TIMES (in-kernel wall_clock64 stamps + host clock stamps of the host-driven step), for
#   VBMC_HIP_LIB=$PWD/variants/libvbmc_st.so python tools/step_times.py [config]
#   VBMC_HIP_LIB=$PWD/variants/libvbmc_st.so python tools/fin_times.py
set -e
cd "$(dirname "$0")/.."
python -m pyvbmc_amd.build > /dev/null
mkdir -p variants
OBJ=pyvbmc_amd/csrc/_obj
for f in prep entropy api_elbo; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DFIN_TIMES \
    -c pyvbmc_amd/csrc/$f.hip -o variants/${f}_st.o
done
objs=$(ls $OBJ/*.o | grep -v "/prep.o\|/entropy.o\|/api_elbo.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/prep_st.o variants/entropy_st.o variants/api_elbo_st.o \
  -o variants/libvbmc_st.so -ldl -Wl,-rpath,/opt/rocm/lib
echo variants/libvbmc_st.so
