#!/usr/bin/env python3
"""For every kernel of a csrc/*.hip file: how many global loads are followed almost immediately (<= 3 instructions) by an
s_waitcnt that drains them — the signature of fetches serialised by divergent branches (round 5: the DIN row gathers).
python tools/isa_load_waits.py paddlerec_amd/csrc/deepfm_fm.hip [name-substring]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/_isa_%d.s" % os.getpid()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"),
                "-I" + os.path.join(REPO, "paddlerec_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
os.remove(out)
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name or "kernel" not in name:
        continue
    ins = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    loads = [i for i, l in enumerate(ins) if l.startswith(("global_load", "buffer_load", "flat_load"))]
    hot = 0
    for i in loads:
        for l in ins[i + 1:i + 4]:
            if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                hot += 1
                break
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%4d loads, %3d waited on at once, %5d instr  %s" % (len(loads), hot, len(ins), demangled[:110]))
