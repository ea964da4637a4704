"""Host-side spans around C-ABI calls (the role START_COMPUTE_SPAN plays in the reference,
cpp/src/arrow/util/tracing_internal.h:141-232; one span per FunctionExecutorImpl::Execute,
function.cc:224-231).  With a KernelTimer installed every span is bracketed by HIP events on
the stream the kernels are enqueued on, which is how bench.py measures the dominant kernel live."""
from __future__ import annotations

import contextlib

import torch

_timer = None


class KernelTimer:
    """Collects (start, end) event pairs per span name; elapsed times are read after a sync."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.events: dict[str, list] = {}

    def start(self, name: str):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream(self.device))
        self.events.setdefault(name, []).append((s, e))
        return e

    def stop(self, end_event):
        end_event.record(torch.cuda.current_stream(self.device))

    def reset(self):
        self.events.clear()

    def elapsed_ms(self, name: str):
        """List of per-span durations in ms (call after torch.cuda.synchronize())."""
        return [s.elapsed_time(e) for s, e in self.events.get(name, [])]


def install(timer: KernelTimer | None) -> None:
    global _timer
    _timer = timer


@contextlib.contextmanager
def span(name: str):
    t = _timer
    if t is None:
        yield
        return
    e = t.start(name)
    try:
        yield
    finally:
        t.stop(e)
