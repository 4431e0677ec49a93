def start_profiler(*a, **k):
    raise NotImplementedError("paddle.utils.profiler: profile with rocprofv3 (tools/pmc.sh, tools/profile_bench.sh)")


def stop_profiler(*a, **k):
    raise NotImplementedError("paddle.utils.profiler: profile with rocprofv3")
