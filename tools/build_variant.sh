#!/bin/bash
# Kernel experiments: tools/build_variant.sh NAME FILE.hip "-DFLAG ..." builds gpurun_tmp/variants/NAME.so = the product library with
# FILE.hip recompiled under the extra flags (the other objects are the in-tree ones).  Use with RA_LIB_PATH=... (see _lib.py).
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/relightableavatar_amd/csrc
name=$1; src=$2; flags=$3
mkdir -p $R/gpurun_tmp/variants
make -s -j8 -C $C
base=$(basename $(basename $src .hip) .cpp)
nos=-fno-slp-vectorize; case $base in ra_k3cc_*) nos="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1";; esac      # no SLP-packed fp32 anywhere (csrc/Makefile)
ext=hip; xf=""; [ -f $C/$base.cpp ] && { ext=cpp; xf="-x hip"; }
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $nos $flags $xf -c $C/$base.$ext -o $R/gpurun_tmp/variants/$name.o
objs=""
for o in ra_k3_f16 ra_k3_bf16 ra_k3c_f16 ra_k3cc_f16 $EXTRA_OBJ ra_k4_fwd_f16 ra_k4_bwd_f16 ra_k4_fwd_bf16 ra_k4_bwd_bf16 ra_hdq ra_trace ra_image ra_api ra_pack ra_shard; do
  ov=""; for kv in $OVERRIDE; do [ "${kv%%=*}" = $o ] && ov="${kv#*=}"; done       # OVERRIDE="ra_k3_f16=path.o ...": further objects to swap
  if [ $o = $base ]; then objs="$objs $R/gpurun_tmp/variants/$name.o"; elif [ -n "$ov" ]; then objs="$objs $ov"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/gpurun_tmp/variants/$name.so
echo built gpurun_tmp/variants/$name.so
