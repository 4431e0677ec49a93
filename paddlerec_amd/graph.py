"""hipGraph replay of a launch-bound training step.

At the reference's own batch sizes (din/config.yaml: batch_size 32; dcn_v2 / deepfm configs: 2..512) a train step is
~60 kernel launches of a few microseconds each: the host's launch path, not the GPU, sets the step time (DIN B 32,
T 152: 0.70 ms eager).  `StepGraph` records the step's launches ONCE per input signature into a hipGraph
(torch.cuda.CUDAGraph is hipGraph on ROCm; the engine's C-ABI launches on torch's current stream, so they are
captured like torch's own kernels) and replays it with the new batch copied into the graph's static input buffers.

Rules the wrapped step must obey (the engine's C-ABI does by construction: no allocation, no host sync, data-dependent
sizes stay on the device behind capacity-sized launches):
  * no host read-back, no python control flow on device values;
  * every host scalar that changes from step to step (learning rate, Adam's step count) must be part of `key`:
    it is baked into the captured kernel arguments.  SGD models qualify as they are; Adam's bias correction changes
    every step, so Adam steps are not graphed;
  * the outputs are STATIC tensors, overwritten by the next replay: consume (or clone) them before stepping again.

The owner of a StepGraph must not form a reference cycle with it (hand in a function that reaches the model through a
weakref, as DINLayer.train_step_graphed does): a model dropped by its last reference then frees its graphs at once.

No step is executed twice and none is executed "for warm-up": the first call with a signature runs eagerly (it also
creates the step's lazily-allocated buffers), the second call is captured — capture records, it does not execute —
and then replayed once, every later call is a replay.
"""
import collections
import gc

import torch


class StepGraph:
    def __init__(self, fn, state_factory=None, max_graphs=32):
        """fn(state, *tensors, **consts) -> tensor or tuple of tensors.  state = state_factory() once per signature:
        the step's reusable buffers (workspaces, saved activations, grouping outputs).  A graph holds the ADDRESSES
        of everything its kernels touch, so buffers that a step with another shape would grow or replace must not be
        shared between signatures — each signature owns its set, alive as long as its graph.
        max_graphs: signatures kept (least recently used evicted; variable-length batches produce one signature per
        padded length)."""
        self.fn = fn
        self.state_factory = state_factory or (lambda: None)
        self.max_graphs = max_graphs
        self._seen = {}
        self._graphs = collections.OrderedDict()
        self.replays = 0
        self.captures = 0
        self.eager = 0

    @staticmethod
    def _signature(tensors, consts):
        return (tuple((tuple(t.shape), t.dtype, t.stride()) for t in tensors), tuple(sorted(consts.items())))

    def __call__(self, *tensors, **consts):
        if not tensors or not tensors[0].is_cuda:          # CPU operator backend (host-logic tests): plain call
            self.eager += 1
            if None not in self._seen:
                self._seen[None] = self.state_factory()
            return self.fn(self._seen[None], *tensors, **consts)
        key = self._signature(tensors, consts)
        entry = self._graphs.get(key)
        if entry is not None:
            graph, static_in, out, _state = entry
            self._graphs.move_to_end(key)
            for dst, src in zip(static_in, tensors):
                dst.copy_(src, non_blocking=True)
            graph.replay()
            self.replays += 1
            return out
        if key not in self._seen:                            # first sight: eager (sizes the signature's buffers)
            if len(self._seen) >= 4 * self.max_graphs:       # signatures seen once and never again: forget the oldest
                self._seen.pop(next(iter(self._seen)))
            state = self._seen[key] = self.state_factory()
            self.eager += 1
            return self.fn(state, *tensors, **consts)
        state = self._seen.pop(key)
        static_in = [t.clone() for t in tensors]
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        # Nothing may destroy a HIP object while the stream captures (hipGraphDestroy / hipEventDestroy are "not
        # permitted when stream is capturing" and abort the process from a destructor): dead python cycles that own
        # graphs, events or streams are collected NOW, and the cyclic collector stays off until the capture has ended.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(graph):
                out = self.fn(state, *static_in, **consts)
        finally:
            if gc_was_on:
                gc.enable()
        self.captures += 1
        if len(self._graphs) >= self.max_graphs:
            self._graphs.popitem(last=False)
        self._graphs[key] = (graph, static_in, out, state)
        graph.replay()                                       # the capture recorded the step; this executes it
        self.replays += 1
        return out
