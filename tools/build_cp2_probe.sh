#!/bin/bash
# Diagnostic library for tools/cp2_probe.py: libllv.so with conv_igemm.hip rebuilt with -DLLV_CP2_STAMPS (phase stamps in the fused
# CvBlock kernels). Not the product: lives under tools/_build/, never loaded by the package.
set -e
cd "$(dirname "$0")/../lowlightenvironmentvideoobjectdetection_amd/csrc"
make -j4 >/dev/null
mkdir -p ../../tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -DLLV_CP2_STAMPS -c conv_igemm.hip -o ../../tools/_build/conv_igemm_stamps.o
OBJS=$(ls build/*.o | grep -v conv_igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_build/libllv_cp2probe.so ../../tools/_build/conv_igemm_stamps.o $OBJS
echo built tools/_build/libllv_cp2probe.so
