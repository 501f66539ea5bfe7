#!/bin/bash
# build_variant.sh <name> <extra compiler flags...>: an ablation / A-B build of the library into variants/lib_<name>.so (bench.py and the
# tests load it with FHESI_LIB=...); the objects go to a scratch directory so the regular build is untouched.
set -e
name=$1; shift
d=variants/obj_$name; mkdir -p $d
for f in hostmath.cpp kernels_ntt.hip kernels_ew.hip kernels_crt.hip kernels_ct.hip kernels_sample.hip kernels_ksaux.hip kernels_aux32.hip kernels_tensor32.hip bluestein.hip comm.hip capi_ctx.hip capi_dcrt.hip capi_pipeline.hip capi_ct.hip; do
  o=$d/${f%.*}.o
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value "$@" -x hip -c $f -o $o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o variants/lib_$name.so $d/*.o -ldl -lpthread
echo built variants/lib_$name.so
