"""paddle.static: only what the dygraph entry points import at module load (tools/utils/save_load.py names
paddle.static.* inside functions the dygraph trainer never calls).  The static-graph executor itself is outside the
engine's scope."""


def data(*a, **k):
    raise NotImplementedError("paddle.static.data: the static-graph path is not part of the engine")
