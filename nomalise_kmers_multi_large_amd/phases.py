"""Per-phase timing of a multi-GPU step (bench and test plumbing): HIP events on the current stream around the device
phases, the host clock around the host round trips, byte counters for what crosses the links."""
import time

import torch


class Phases:
    """phases.dev("count") / phases.host("sizes") are context managers; nothing is recorded when disabled.
    A device phase is the time between two events on torch's current stream (the library launches on that stream), so a
    phase that waits for a collective shows the part of it that nothing else covered."""

    def __init__(self, enabled=False, device=None):
        self.enabled = bool(enabled) and torch.cuda.is_available() and device is not None and str(device) != "cpu"
        self.host_only = bool(enabled) and not self.enabled
        self.device = device
        self._open = []      # (name, start event, stop event)
        self.ms = {}
        self.bytes = {}
        self.steps = 0

    class _Dev:
        def __init__(self, p, name):
            self.p, self.name = p, name

        def __enter__(self):
            if self.p.enabled:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            elif self.p.host_only:
                self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            if self.p.enabled:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.p._open.append((self.name, self.e0, e1))
            elif self.p.host_only:
                self.p.ms[self.name] = self.p.ms.get(self.name, 0.0) + (time.perf_counter() - self.t0) * 1e3

    class _Host:
        def __init__(self, p, name):
            self.p, self.name = p, name

        def __enter__(self):
            if self.p.enabled or self.p.host_only:
                self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            if self.p.enabled or self.p.host_only:
                self.p.ms[self.name] = self.p.ms.get(self.name, 0.0) + (time.perf_counter() - self.t0) * 1e3

    def dev(self, name):
        return Phases._Dev(self, name)

    def host(self, name):
        return Phases._Host(self, name)

    def add_bytes(self, name, n):
        if self.enabled or self.host_only:
            self.bytes[name] = self.bytes.get(name, 0) + int(n)

    def step_done(self):
        self.steps += 1
        if len(self._open) > 4096:
            self.collect()

    def collect(self):
        if self._open:
            torch.cuda.synchronize(self.device)
            for name, e0, e1 in self._open:
                self.ms[name] = self.ms.get(name, 0.0) + e0.elapsed_time(e1)
            self._open = []

    def reset(self):
        self.collect()
        self.ms, self.bytes, self.steps = {}, {}, 0

    def report(self):
        """{"steps", "ms_per_step": {phase: ms}, "bytes_per_step": {what: bytes}}"""
        self.collect()
        n = max(1, self.steps)
        return {"steps": self.steps, "ms_per_step": {k: v / n for k, v in self.ms.items()},
                "bytes_per_step": {k: v / n for k, v in self.bytes.items()}}
