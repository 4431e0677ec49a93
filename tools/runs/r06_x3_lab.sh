#!/bin/bash
# builds lab variants of librecengine.so IN THE BUILD CONTAINER (call with: build <n>) or times them on the GPU box (run)
if [ "$1" = build ]; then
  for v in 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ipaddlerec_amd/csrc -Wno-unused-result -DREC_X3_LAB=$v -c paddlerec_amd/csrc/gemm_f32.hip -o /tmp/gemm_lab$v.o &
  done; wait
  mkdir -p gpurun_lab
  for v in 1 2 3 4; do
    objs=$(ls paddlerec_amd/_obj/*.o | grep -v gemm_f32.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_lab/librecengine_lab$v.so $objs /tmp/gemm_lab$v.o
  done
  ls -la gpurun_lab
else
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
  echo "== shipped"; REC_X3_WN=1 python tools/x3_wn_bench.py 2>&1 | grep -v amdgpu | head -4
  for v in 1 2 3 4; do
    cp paddlerec_amd/librecengine.so /tmp/keep.so; cp gpurun_lab/librecengine_lab$v.so paddlerec_amd/librecengine.so
    echo "== lab $v (1 no C stores, 2 no A loads, 3 no LDS-DMA, 4 no conversion)"; python tools/x3_wn_bench.py 2>&1 | grep -v amdgpu | head -4
    cp /tmp/keep.so paddlerec_amd/librecengine.so
  done
fi
