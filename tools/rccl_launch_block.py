#!/usr/bin/env python3
"""Does a kernel launch that follows an RCCL collective on a stream block the HOST until the GPU gets there?
(world 1, nccl backend; rec_stream_spin provides the GPU-side delay).  Prints host times of the launch."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from paddlerec_amd._lib import lib

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
L = lib()
spin = lambda us, st: L.rec_stream_spin(us, C.c_void_p(st.cuda_stream))
x = torch.ones(1 << 20, device=dev)
y = torch.empty_like(x)
dist.all_to_all_single(y, x)            # communicator set-up
torch.cuda.synchronize()


def case(name, fn):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        ts.append(fn())
        torch.cuda.synchronize()
    print("%-58s host ms of the probed call(s): %s" % (name, " ".join("%.3f" % (1e3 * t) for t in ts)))


s = torch.cuda.Stream(device=dev)
s2 = torch.cuda.Stream(device=dev)


def timed(f):
    t0 = time.perf_counter()
    f()
    return time.perf_counter() - t0


def a():
    with torch.cuda.stream(s):
        spin(2000, s)
        dist.all_to_all_single(y, x)
        return timed(lambda: spin(1, s))


def b():
    with torch.cuda.stream(s):
        spin(2000, s)
        dist.all_to_all_single(y, x)
        e = torch.cuda.Event()
        e.record(s)
    s2.wait_event(e)
    return timed(lambda: spin(1, s2))


def c():
    with torch.cuda.stream(s):
        spin(2000, s)
        return timed(lambda: spin(1, s))


def d():
    with torch.cuda.stream(s):
        spin(2000, s)
        w = dist.all_to_all_single(y, x, async_op=True)
        w.wait()
        return timed(lambda: spin(1, s))


def e_():
    with torch.cuda.stream(s):
        spin(2000, s)
        return timed(lambda: dist.all_to_all_single(y, x))


def f():
    with torch.cuda.stream(s):
        spin(2000, s)
        dist.all_to_all_single(y, x)
        return timed(lambda: dist.all_to_all_single(y, x))


def g():
    with torch.cuda.stream(s):
        spin(2000, s)
        dist.all_to_all_single(y, x)
        return timed(lambda: y.add_(1.0))


def h():
    with torch.cuda.stream(s):
        spin(2000, s)
        dist.all_reduce(x)
        return timed(lambda: spin(1, s))


case("control: spin, then launch on the same stream", c)
case("spin, all_to_all, then launch on the same stream", a)
case("spin, all_to_all, then launch on another stream (event)", b)
case("spin, all_to_all(async_op)+wait, then launch", d)
case("spin, then the all_to_all call itself", e_)
case("spin, all_to_all, then a second all_to_all", f)
case("spin, all_to_all, then a torch kernel on the same stream", g)
case("spin, all_reduce, then launch on the same stream", h)
dist.destroy_process_group()
