"""Frames in flight: the throughput mode of the render path (no counterpart in the reference, whose frame loop is sequential:
lib/evaluators/*, run.py).

One frame of this path is two very different phases: a chain of ~40 small, latency-bound launches (pose, box structure, the 16
surface-tracing iterations, normals, shading: about 2 ms at any frame size) and the light-visibility stage, whose few large launches
fill the chip.  A `FramePipeline` keeps `depth` frames in flight on `depth` HIP streams, each with its own context (weights, frame
state, scratch); the contexts share a `Gate`, so the large stages run one after the other in submission order and the small phase
of frame f + 1 runs beside the stage of frame f.  Frames are bit-identical to sequential rendering (tests/test_gpu_parity.py).
Depth 3 is the measured optimum (one frame in its stage, two in their small phases: 512 x 512 relight 31.1 / 29.2 / 28.9 / 29.2 ms at
depth 1 / 2 / 3 / 4, one rank of eight 6.8 / 4.8 / 4.25 / 4.9 ms).

    pipe = FramePipeline(cfg, state_dict, device, depth=3)
    for batch in loader:
        pending.append(pipe.submit(batch))            # returns at once: everything is queued on the replica's stream
    for p in pending:
        out = p.result()                              # makes the caller's current stream wait for that frame

A batch must stay UNCHANGED until its frame has been consumed (`ra_set_frame` reads R / Th / pnorm / tverts in place); the pipeline keeps it
alive (the Pending holds it, and its tensors are recorded on the replica's stream, so dropping the batch right after submit() is safe).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check


class Gate:
    """ra_gate of the C ABI (include/relightableavatar.h)"""

    def __init__(self, device):
        self.lib = _lib.lib()
        self.handle = C.c_void_p()
        dev = torch.device(device)
        check(self.lib.ra_gate_create(C.byref(self.handle), dev.index or 0), 'ra_gate_create')

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.handle.value:
                self.lib.ra_gate_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


def _record(v, stream):
    if isinstance(v, torch.Tensor):
        if v.is_cuda:
            v.record_stream(stream)
    elif isinstance(v, dict):
        for x in dict.values(v):            # not v.values(): a lazydict would evaluate its lazy entries
            _record(x, stream)
        if hasattr(v, 'pending_tensors'):   # what its unevaluated thunks close over (allocated on the replica's stream, read on the consumer's)
            _record(v.pending_tensors(), stream)
    elif isinstance(v, (list, tuple)):
        for x in v:
            _record(x, stream)


class Pending:
    """a frame queued on a replica's stream"""

    def __init__(self, value, event, stream, inputs=None):
        self._value, self._event, self.stream = value, event, stream
        self._inputs = inputs           # the batch stays referenced at least as long as its frame is pending

    def result(self):
        """the frame's output; the caller's current stream waits for it (no host synchronisation)"""
        cur = torch.cuda.current_stream(self.stream.device)
        if cur != self.stream:
            cur.wait_event(self._event)
            _record(self._value, cur)       # allocated on the replica's stream, consumed on the caller's: tell the caching allocator
        return self._value

    def synchronize(self):
        self._event.synchronize()
        return self._value


class FramePipeline:
    def __init__(self, cfg, state_dict, device, depth=3):
        from .networks import make_network
        from .renderer import make_renderer
        assert depth >= 1
        self.device = torch.device(device)
        self.depth = depth
        self.gate = Gate(self.device) if depth > 1 else None
        self.networks, self.renderers, self.streams = [], [], []
        for _ in range(depth):
            net = make_network(cfg)
            net.load_state_dict(state_dict)
            net = net.to(self.device).eval()
            if self.gate is not None:
                net.engine().set_gate(self.gate)
            self.networks.append(net)
            self.renderers.append(make_renderer(cfg, net))
            # depth 1: the caller's stream (plain sequential rendering)
            self.streams.append(torch.cuda.Stream(self.device) if depth > 1 else None)
        self._next = 0

    def submit(self, batch=None, fn=None):
        """queue one frame on the next replica: `renderer.render(batch)`, or `fn(network, renderer)` for callers that shard / gather /
        post-process on the same stream.  Everything `fn` launches must go to the current stream."""
        r = self._next
        self._next = (r + 1) % self.depth
        net, rend, st = self.networks[r], self.renderers[r], self.streams[r]
        if st is None:
            st = torch.cuda.current_stream(self.device)
        else:
            st.wait_stream(torch.cuda.current_stream(self.device))     # the inputs were produced on the caller's stream
            # the batch was allocated on the caller's stream and is read on the replica's (the library also keeps POINTERS into it until
            # the frame is done: ra_set_frame reads R / Th / pnorm / tverts in place): tell the caching allocator, or memory the caller
            # frees right after submit() is handed out again while the frame still reads it
            _record(batch, st)
        with torch.cuda.stream(st):
            val = fn(net, rend) if fn is not None else rend.render(batch)
            ev = torch.cuda.Event()
            ev.record(st)
        return Pending(val, ev, st, batch)

    def engines(self):
        return [n.engine() for n in self.networks]
