#!/bin/bash
# builds libmvlt_hip.so with -DG8_TRACE into a scratch copy of the package and runs scripts/g8_trace.py against it
set -e
R=$GRAFT_REPO_ROOT; S=/tmp/g8trace; rm -rf $S; mkdir -p $S
cp -r $R/medical-vision-langauge-transformer_amd $R/mvlt_amd $R/include $R/scripts $S/
cd $S/medical-vision-langauge-transformer_amd/csrc
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -DG8_TRACE -c gemm8.hip -o gemm8.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmvlt_hip.so gemm.o gemm8.o rowstream.o norm.o attn.o misc.o wmsa.o wmsa2.o
cd $S && python scripts/g8_trace.py
