"""Round 6: what an event record between two LARGE kernels costs, by event flags.  A default HIP event performs a
system-scope fence when it is recorded; hipEventDisableSystemFence / hipEventReleaseToDevice events do not.
main: big, record(e), big   with a consumer on another queue (side: wait(e), small kernel) — per pair, against plain."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
cur = torch.cuda.current_stream(); side = ops.concurrent_stream(dev)
x = torch.zeros(64 << 20, device=dev)          # 256 MB: ~100 us per in-place pass, every line dirty
y = torch.zeros(1 << 16, device=dev)
def big(): x.add_(1.0)
def small(st):
    with torch.cuda.stream(st): y.add_(1.0)
FLAGS = {"default (DisableTiming)": 0x2, "DisableSystemFence": 0x2 | 0x20000000, "ReleaseToDevice": 0x2 | 0x40000000,
         "ReleaseToSystem": 0x2 | 0x80000000}
N = 100
def timed(fn):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(N): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / N
print("plain big, big: %.1f us" % timed(lambda: (big(), big())))
for name, fl in FLAGS.items():
    ev = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(ev), C.c_uint(fl)) == 0
    def rec_only():
        big(); hip.hipEventRecord(ev, C.c_void_p(cur.cuda_stream)); big()
    def fork():
        big(); hip.hipEventRecord(ev, C.c_void_p(cur.cuda_stream)); hip.hipStreamWaitEvent(C.c_void_p(side.cuda_stream), ev, 0); small(side); big()
    def join():
        small(side); hip.hipEventRecord(ev, C.c_void_p(side.cuda_stream)); big(); hip.hipStreamWaitEvent(C.c_void_p(cur.cuda_stream), ev, 0); big()
    print("%-26s record only %.1f   fork (consumer on side) %.1f   join (wait fired event of side) %.1f us" % (name, timed(rec_only), timed(fork), timed(join)))
