"""Register budgets of the roofline kernels, checked at build time (no GPU needed: hipcc cross-compiles gfx950 and reports
each kernel's resource usage).  The dense-layout FM backward must keep 3 waves per SIMD and the FM forward 4: round 4 lost
one wave of the backward to a harmless-looking extra kernel argument (164 -> 180 VGPRs, 65 -> 77 us, the bench's roofline
fraction 0.656 -> 0.597) and only the bench line showed it."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_fm_kernels_keep_their_occupancy(tmp_path):
    src = os.path.join(REPO, "paddlerec_amd", "csrc", "deepfm_fm.hip")
    saved = os.path.join(REPO, "paddlerec_amd", "_obj", "deepfm_fm.resources.txt")    # written by paddlerec_amd.build
    deps = [src, os.path.join(REPO, "paddlerec_amd", "csrc", "fm_tile.h"), os.path.join(REPO, "paddlerec_amd", "csrc", "rec_common.h")]
    if os.path.exists(saved) and all(os.path.getmtime(d) <= os.path.getmtime(saved) for d in deps) \
            and "Occupancy" in open(saved).read():
        _check_occupancy(open(saved).read())
        return
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"),
                        "-I" + os.path.join(REPO, "paddlerec_amd", "csrc"), "-c", src, "-o", str(tmp_path / "fm.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_occupancy(r.stderr)


def _check_occupancy(remarks):
    occ, name = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
        if m and name:
            occ[name] = int(m.group(1))
    # fm_bwd_kernel<VEC 4, LANES 4, NDI 2, NT, PADDED false> (D 16, 13 dense fields: BASELINE configs[1]), both NT forms
    bwd = [v for k, v in occ.items() if "fm_bwd_kernelILi4ELi4ELi2E" in k and k.split("fm_bwd_kernelILi4ELi4ELi2E")[1].startswith(("Lb1ELb0E", "Lb0ELb0E"))]
    fwd = [v for k, v in occ.items() if "fm_fwd_kernelILi4ELi4ELi1E" in k]
    assert bwd and fwd, sorted(occ)[:10]
    assert min(bwd) >= 3, "fm_bwd_kernel<4,4,2,*,false> fell to %d waves per SIMD" % min(bwd)
    assert min(fwd) >= 4, "fm_fwd_kernel<4,4,1,*> fell to %d waves per SIMD" % min(fwd)
    # the block-tile kernels for narrow rows (fm_tile.h): 6 blocks of 256 threads per CU is what their 16-sample tiles are
    # sized for (round 5: 117 VGPRs = 4 waves per SIMD without the launch bound, 64 -> 52 us with it)
    tile_f = [v for k, v in occ.items() if "fm_fwd_tile_kernel" in k]
    tile_b = [v for k, v in occ.items() if "fm_bwd_tile_kernel" in k]
    assert len(tile_f) == 4 and len(tile_b) >= 3, sorted(occ)
    assert min(tile_f) >= 6 and min(tile_b) >= 6, (tile_f, tile_b)


def test_bf16x3_gemm_kernels_keep_their_registers():
    """csrc/gemm_bf16x3.h: the eight-wave forward / dX kernel lives on TWO waves per SIMD (<= 256 registers; it is 7-8 %
    slower with one), none of the kernels may spill (the weight-gradient kernel sits at 440 of 512 registers)."""
    saved = os.path.join(REPO, "paddlerec_amd", "_obj", "gemm_f32.resources.txt")
    src = [os.path.join(REPO, "paddlerec_amd", "csrc", f) for f in ("gemm_f32.hip", "gemm_bf16x3.h", "gemm_epi.h")]
    if not (os.path.exists(saved) and all(os.path.getmtime(d) <= os.path.getmtime(saved) for d in src)):
        pytest.skip("no resource remarks newer than the sources (python -m paddlerec_amd.build writes them)")
    occ, scratch, name = {}, {}, None
    for line in open(saved):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
        if m and name:
            occ[name] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    x3 = {k: v for k, v in occ.items() if "gemm_bf16x3" in k}
    assert len(x3) >= 30, sorted(x3)[:5]
    # gemm_bf16x3_kernel<NT, EPI, WM 4, WN>: the four MLP epilogues at every column-block width, in both workgroup shapes
    # (WN 1, round 6: four waves, two workgroups per CU; WN 2, round 5: eight waves in one) — two waves per SIMD either way
    eight = [k for k in x3 if re.search(r"gemm_bf16x3_kernelILi(13|8|7)ELi[0-3]ELi4ELi[12]E", k)]
    assert len(eight) == 24 and all(x3[k] >= 2 for k in eight), {k: x3[k] for k in eight}
    assert all(scratch[k] == 0 for k in x3 if "dw_kernel" in k or k in eight), {k: scratch[k] for k in x3 if scratch[k]}
