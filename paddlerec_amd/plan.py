"""Recorded call list of a launch-bound training step.

At the reference's own batch sizes (deepfm / dcn_v2 `config_bigdata.yaml`: 512, din: 32) a train step is ~40 C-ABI calls
of a few microseconds of GPU work each, and ~9 us of python + ctypes argument marshalling per call set the step time
(DeepFM B 512: 0.35 ms of host time for 0.25 ms of GPU time, profiles/r03_small_batch.txt).  In the steady state those
calls are IDENTICAL from step to step — same functions, same buffers, same sizes — except for the input pointers and
Adam's step count.  `CallPlan` records them once (every `lib().rec_*` call an `ops` wrapper makes, with the ctypes
argument objects it built) and replays the list: `fn(*args)` per call, the input pointers and the `rec_adam_hyper`
structs patched in place.  This is the host half of what one "issue the whole step" C entry point would do, without
stating the step's orchestration a second time in C++; a hipGraph of the same step replays no faster than the eager
step on this ROCm, because the GPU-side cost per dependent kernel is what it is — the plan only
removes the host's share.

Rules for a recorded step (the mirrors obey them in their small-batch path): C-ABI calls only (a torch kernel would not
be replayed — tests compare planned and eager steps bit for bit over several steps), one stream, no host read-back, no
python control flow on device values; every tensor a call touches is kept alive by the plan."""
import ctypes as C

import torch

from . import _lib, ops

_HOST_ONLY = ("rec_last_error", "rec_gemm_plan_splits", "rec_din_saves_act1", "rec_comm_available")


class CallPlan:
    def __init__(self):
        self.calls = []          # (function, args tuple)
        self.keep = []           # tensors whose addresses the calls hold
        self.pointers = []       # (c_void_p object, address at record time)
        self.hypers = []         # rec_adam_hyper structs passed by reference
        self.input_slots = []    # (c_void_p object, input index)
        self.fields = []         # (struct, field name, address at record time): pointers inside descriptor arrays
        self.field_slots = []    # (struct, field name, input index)
        self.step_args = []      # (c_uint64 object, value at record time, increment per step): counters derived from the step
        self.step0 = 0
        self.stream = None
        self.outputs = None
        self.output_slots = []   # (c_void_p object, output index): where the recorded step wrote its returned tensors
        self.output_fields = []  # (struct, field name, output index)

    # -- recording ---------------------------------------------------------------------------------
    def note_pointer(self, p, t):
        self.keep.append(t)
        self.pointers.append((p, t.data_ptr()))

    def note_step_arg(self, obj, stride):
        """obj (a ctypes integer passed by value) = a + stride * step for some a: re-derived on every replay."""
        self.step_args.append((obj, int(obj.value), int(stride)))

    def note_field(self, obj, field, t):
        self.keep.append(t)
        self.fields.append((obj, field, t.data_ptr()))

    def proxy(self, h):
        plan = self

        class _Proxy:
            def __getattr__(self, name):
                fn = getattr(h, name)
                if name.endswith("_bytes") or name in _HOST_ONLY:      # host-side size queries: not part of the step
                    return fn

                def call(*args):
                    plan.calls.append((fn, args))
                    for a in args:
                        obj = getattr(a, "_obj", None)                 # C.byref(struct)
                        if isinstance(obj, _lib.AdamHyper):
                            plan.hypers.append(obj)
                    return fn(*args)
                return call
        return _Proxy()

    def record(self, fn, inputs, step0=0):
        """Run fn() with every C-ABI call listed; inputs: the tensors whose pointers change from step to step; step0: the
        step count of the recorded step (for arguments registered with note_step_arg)."""
        self.step0 = int(step0)
        if ops._recorder is not None:
            raise ops.RecError("a step is already being recorded")
        self.stream = ops._stream().value
        ops._recorder = self
        try:
            self.outputs = fn()
        finally:
            ops._recorder = None
        addr = {t.data_ptr(): i for i, t in enumerate(inputs)}
        self.input_slots = [(p, addr[a]) for p, a in self.pointers if a in addr]
        self.field_slots = [(o, f, addr[a]) for o, f, a in self.fields if a in addr]
        # the tensors the step RETURNS (loss, pred): a replay writes them into fresh tensors, as the eager step does — a
        # caller that keeps per-step predictions must not see every later step overwrite them
        outs = self.outputs if isinstance(self.outputs, (tuple, list)) else (self.outputs,)
        oaddr = {t.data_ptr(): i for i, t in enumerate(outs) if torch.is_tensor(t) and t.data_ptr() not in addr}
        self.output_slots = [(p, oaddr[a]) for p, a in self.pointers if a in oaddr]
        self.output_fields = [(o, f, oaddr[a]) for o, f, a in self.fields if a in oaddr]
        self.fields = None
        self.input_sig = [(tuple(t.shape), t.dtype, t.stride()) for t in inputs]
        self.pointers = None
        return self.outputs

    # -- replay ------------------------------------------------------------------------------------
    def matches(self, inputs):
        return (ops._stream().value == self.stream and
                [(tuple(t.shape), t.dtype, t.stride()) for t in inputs] == self.input_sig)

    def replay(self, inputs, step, lr):
        for p, i in self.input_slots:
            p.value = inputs[i].data_ptr()
        for o, f, i in self.field_slots:
            setattr(o, f, inputs[i].data_ptr())
        for h in self.hypers:
            h.step = step
            h.lr = lr
        for o, base, stride in self.step_args:
            o.value = base + (step - self.step0) * stride
        outs = self.outputs
        if self.output_slots or self.output_fields:
            single = not isinstance(outs, (tuple, list))
            fresh = [torch.empty_like(t) if torch.is_tensor(t) else t for t in ((outs,) if single else outs)]
            for p, i in self.output_slots:
                p.value = fresh[i].data_ptr()
            for o, f, i in self.output_fields:
                setattr(o, f, fresh[i].data_ptr())
            outs = fresh[0] if single else tuple(fresh)
        for fn, args in self.calls:
            rc = fn(*args)
            if rc:
                _lib.check(rc, "replayed call")
        return outs
