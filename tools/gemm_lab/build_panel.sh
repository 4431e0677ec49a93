#!/bin/bash
# builds tools/gemm_lab/_build/libpanellab.so (csrc/gemm_panel.h alone); extra args go to hipcc (-DREC_PANEL_...=...)
cd "$(dirname "$0")/../.." && mkdir -p tools/gemm_lab/_build && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipaddlerec_amd/csrc -fPIC -shared "$@" \
  tools/gemm_lab/panel_lab.hip -o tools/gemm_lab/_build/libpanellab${SUFFIX}.so
