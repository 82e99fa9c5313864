#!/bin/bash
# Build a variant of the library with extra compiler flags: scripts/exp/build_variant.sh NAME -DFLAG ...  -> build/variants/libsaga_NAME.so
# (load it with SAGA_HIP_LIB=build/variants/libsaga_NAME.so; build/ travels to the GPU box, it is only git-ignored)
set -e
fail=0
NAME=$1; shift
cd "$(dirname "$0")/../../saga_amd/csrc"
OUT=../../build/variants; OBJ=../../build/obj_$NAME
mkdir -p $OUT $OBJ
# the experiment switches only compile in an experiment build, whose saga_version() is negative: load with SAGA_ALLOW_EXPERIMENT_LIB=1
EXP=""; case "$*" in *_EXP_*|*GL_PHASE_TIMING*) EXP="-DSAGA_EXPERIMENT_BUILD";; esac
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast-honor-pragmas $EXP $*"
for f in model pose blend_gemm skin chamfer normals smplx_api grasp_loss rot6d fit_step rows motion_terms motion_step; do
  X=""; [ $f = blend_gemm ] && X="-mllvm -amdgpu-mfma-vgpr-form=1"      # as the Makefile
  rm -f $OBJ/$f.o; /opt/rocm/bin/hipcc $FLAGS $X -c $f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsaga_$NAME.so $OBJ/*.o
echo built $OUT/libsaga_$NAME.so
