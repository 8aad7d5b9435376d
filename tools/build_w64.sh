#!/bin/bash
# tools/build_w64.sh NAME "FLAGS": a variant of the 64-points-per-wave K3 (csrc/ra_k3w_f16.hip) linked with a ra_k3_f16.o that calls it
set -e
R=$(cd $(dirname $0)/.. && pwd)
[ -f $R/gpurun_tmp/variants/k3_calls_w64.o ] || (cd $R/relightableavatar_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize -DRA_K3_WIDE64 -c ra_k3_f16.hip -o $R/gpurun_tmp/variants/k3_calls_w64.o)
EXTRA_OBJ=ra_k3w_f16 OVERRIDE="ra_k3_f16=$R/gpurun_tmp/variants/k3_calls_w64.o" bash $R/tools/build_variant.sh $1 ra_k3w_f16.hip "$2 -mllvm -amdgpu-mfma-vgpr-form=1 -save-temps=obj"
cp $R/gpurun_tmp/variants/ra_k3w_f16-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/isa/$1.s
rm -f $R/gpurun_tmp/variants/ra_k3w_f16-*
python $R/tools/isa_mix.py /tmp/isa/$1.s stream64 ${3:-14}
