"""What a cross-stream dependency costs on the device timeline (round 6): pairs of short kernels on one stream with
(a) nothing, (b) an event record, (c) a wait on an event of another stream that fired long ago, (d) a wait on an event
that fires right then (the other stream's kernel ends just before) in between.  Device time per pair from HIP events
around 200 pairs; the kernels are rec_stream_spin(20 us) (one wave)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()
cur = torch.cuda.current_stream()
side = ops.concurrent_stream(dev)
spin = lambda us, st: L.rec_stream_spin(int(us), st.cuda_stream)
N = 200

def timed(fn):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(N):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / N

def plain():
    spin(20, cur); spin(20, cur)
def with_record():
    spin(20, cur); torch.cuda.Event().record(cur); spin(20, cur)
old = torch.cuda.Event(); old.record(side); torch.cuda.synchronize()
def wait_fired():
    spin(20, cur); cur.wait_event(old); spin(20, cur)
def wait_fresh_fired():            # a new event per pair, recorded on the idle side stream: fires at once
    e = torch.cuda.Event(); e.record(side)
    spin(20, cur); cur.wait_event(e); spin(20, cur)
def wait_live():                   # the side kernel ends while the first main kernel runs: the wait is satisfied on arrival
    spin(10, side); e = torch.cuda.Event(); e.record(side)
    spin(20, cur); cur.wait_event(e); spin(20, cur)
def wait_late():                   # the side kernel ends 20 us AFTER the first main kernel: a real wait
    spin(40, side); e = torch.cuda.Event(); e.record(side)
    spin(20, cur); cur.wait_event(e); spin(20, cur)
def fork_join():                   # main -> side -> main (what _OnSide + wait_stream does around a side kernel)
    spin(20, cur); side.wait_stream(cur); spin(20, side); cur.wait_stream(side); spin(20, cur)
for name, fn in (("plain (2 x 20 us)", plain), ("event record between", with_record), ("wait on a long-fired event", wait_fired),
                 ("wait on a fresh event of an idle stream", wait_fresh_fired), ("wait, other stream ends early", wait_live),
                 ("wait, other stream ends 20 us late (expect 60)", wait_late), ("fork / join around a side kernel (expect 60)", fork_join)):
    print("%-55s %7.1f us per pair" % (name, timed(fn)))

# --- round 6, second set: does a CONSUMER on another queue slow the producer's queue?
third = ops.concurrent_stream(dev, index=1)
def fork_only():                   # main: k, record, k ; side: wait, k  (never joined back)
    spin(20, cur); e = torch.cuda.Event(); e.record(cur); side.wait_event(e); spin(20, side); spin(20, cur)
def fork_only_big_gap():           # the consumer's kernel is short: does main's second kernel start late?
    spin(20, cur); e = torch.cuda.Event(); e.record(cur); side.wait_event(e); spin(2, side); spin(20, cur)
def join_only():                   # side free-runs ahead; main waits on its (already fired or not) event each pair
    spin(5, side); e = torch.cuda.Event(); e.record(side); spin(20, cur); cur.wait_event(e); spin(20, cur)
def two_consumers():
    spin(20, cur); e = torch.cuda.Event(); e.record(cur); side.wait_event(e); third.wait_event(e); spin(10, side); spin(10, third); spin(20, cur)
for name, fn in (("fork only: main k, record, k | side wait, k (expect 40)", fork_only),
                 ("fork only, short consumer", fork_only_big_gap), ("join only (side ahead)", join_only),
                 ("fork to two queues", two_consumers)):
    print("%-55s %7.1f us per pair" % (name, timed(fn)))
