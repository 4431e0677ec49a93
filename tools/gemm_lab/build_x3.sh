#!/bin/bash
# builds tools/gemm_lab/_build/libx3lab.so (csrc/gemm_bf16x3.h alone); extra args go to hipcc (-DREC_X3_PRODUCTS=3 ...)
cd "$(dirname "$0")/../.." && mkdir -p tools/gemm_lab/_build && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipaddlerec_amd/csrc -fPIC -shared -DREC_X3_LAB_STANDALONE "$@" \
  tools/gemm_lab/bf16x3_lab.hip -o tools/gemm_lab/_build/libx3lab${SUFFIX}.so
