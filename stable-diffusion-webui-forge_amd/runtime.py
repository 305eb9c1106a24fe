"""Device-memory arena and HIP-graph capture for the executors.

MI355X has 288 GB of HBM: weights stay resident and a whole UNet / VAE forward draws its activations from one
pre-allocated arena (stack discipline: mark / release), so (a) nothing calls hipMalloc inside the step loop,
(b) every forward sees the same addresses, which is what lets the forward be captured once into a HIP graph
and replayed (the reference instead re-queries free memory and re-plans batching every step,
backend/sampling/sampling_function.py:193-213).
"""
import ctypes as C

import torch

from . import _lib, hipops


class ArenaOverflow(RuntimeError):
    pass


class Arena:
    ALIGN = 256

    def __init__(self, capacity_bytes, device):
        self.device = device
        self.capacity = int(capacity_bytes)
        self.buf = torch.empty(self.capacity, dtype=torch.uint8, device=device)
        self.top = 0
        self.peak = 0

    def reset(self):
        self.top = 0

    def mark(self):
        return self.top

    def release(self, mark):
        self.top = mark

    def alloc(self, shape, dtype):
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        start = (self.top + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        end = start + nbytes
        if end > self.capacity:
            raise ArenaOverflow(f"arena overflow: need {end} bytes, capacity {self.capacity}")
        self.top = end
        self.peak = max(self.peak, end)
        return self.buf[start:end].view(dtype).view(shape)

    def __enter__(self):
        self._prev = hipops.set_allocator(self.alloc)
        return self

    def __exit__(self, *exc):
        hipops.set_allocator(self._prev)
        return False


class HipGraph:
    """Thin owner of a hipGraphExec_t captured through the C-ABI (fmx_graph_*)."""

    def __init__(self):
        self.exec = None

    def capture(self, stream, fn):
        L = _lib.lib()
        sp = C.c_void_p(stream.cuda_stream)
        _lib.check(L.fmx_graph_begin(sp), "fmx_graph_begin")
        try:
            out = fn()
        except BaseException:
            tmp = C.c_void_p()
            L.fmx_graph_end(sp, C.byref(tmp))
            if tmp.value:
                L.fmx_graph_destroy(tmp)
            raise
        ex = C.c_void_p()
        _lib.check(L.fmx_graph_end(sp, C.byref(ex)), "fmx_graph_end")
        self.destroy()
        self.exec = ex
        return out

    def launch(self, stream):
        _lib.check(_lib.lib().fmx_graph_launch(self.exec, C.c_void_p(stream.cuda_stream)), "fmx_graph_launch")

    def destroy(self):
        if self.exec is not None:
            _lib.lib().fmx_graph_destroy(self.exec)
            self.exec = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipEventTimer:
    """HIP events on an explicit stream (used by bench.py for per-kernel-launch timing)."""

    def __init__(self):
        L = _lib.lib()
        self.a, self.b = C.c_void_p(), C.c_void_p()
        _lib.check(L.fmx_event_create(C.byref(self.a)), "fmx_event_create")
        _lib.check(L.fmx_event_create(C.byref(self.b)), "fmx_event_create")

    def start(self, stream):
        _lib.check(_lib.lib().fmx_event_record(self.a, C.c_void_p(stream.cuda_stream)), "fmx_event_record")

    def stop(self, stream):
        _lib.check(_lib.lib().fmx_event_record(self.b, C.c_void_p(stream.cuda_stream)), "fmx_event_record")

    def elapsed_ms(self):
        ms = C.c_float()
        _lib.check(_lib.lib().fmx_event_elapsed_ms(self.a, self.b, C.byref(ms)), "fmx_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            L = _lib.lib()
            L.fmx_event_destroy(self.a)
            L.fmx_event_destroy(self.b)
        except Exception:
            pass
