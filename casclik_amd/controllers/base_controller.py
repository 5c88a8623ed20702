"""Shared plumbing of the device controllers.

``BaseController`` keeps the reference's ``controller_type<label>`` repr
(reference: casclik/controllers/base_controller.py:1-6).  The helpers move
batches between numpy / torch and the device pointers the C ABI takes; torch
is used purely as the device allocator and stream provider.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class BaseController(object):
    controller_type = "BaseController"

    def __repr__(self):
        return self.controller_type + "<" + self.skill_spec.label + ">"


def _torch():
    import torch
    return torch


def device_of(device=None):
    torch = _torch()
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError(
                "casclik_amd controllers run on an AMD GPU through the HIP "
                "library; no GPU is visible and there is no CPU fallback.")
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        # ("cuda" without an index never compares equal to a tensor's "cuda:0": resolve it once, here)
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def check_out_tensor(t, shape, dtype_name, device, what):
    """An output tensor the kernels write through ``data_ptr()``: it must be exactly what they assume -
    a contiguous torch tensor of that shape and dtype on the controller's device - or the launch would write
    out of bounds / garbage instead of raising."""
    torch = _torch()
    if t is None:
        return
    want = getattr(torch, dtype_name)
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s must be a torch tensor on %s" % (what, device))
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (what, tuple(shape), tuple(t.shape)))
    if t.dtype != want:
        raise ValueError("%s must be %s, got %s" % (what, want, t.dtype))
    if t.device != torch.device(device):
        raise ValueError("%s lives on %s, the controller on %s" % (what, t.device, device))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)


def to_device_matrix(val, width, device, what, batch=None):
    """numpy / list / DM / torch -> contiguous float64 [B, width] tensor on
    ``device``; returns (tensor, was_numpy)."""
    torch = _torch()
    if val is None:
        return None, True
    if isinstance(val, torch.Tensor):
        t = val
        if t.dim() == 1:
            t = t.reshape(1, -1) if width != 1 or t.numel() == 1 else t.reshape(-1, 1)
        t = t.to(device=device, dtype=torch.float64).contiguous()
        was_numpy = False
    else:
        if hasattr(val, "toarray"):
            val = val.toarray()
        arr = np.asarray(val, dtype=np.float64)
        if arr.ndim == 0:
            arr = arr.reshape(1, 1)
        elif arr.ndim == 1:
            arr = arr.reshape(1, -1) if arr.size == width else arr.reshape(-1, width)
        elif arr.ndim == 2 and arr.shape[1] != width and arr.shape[0] == width and arr.shape[1] == 1:
            arr = arr.T
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        was_numpy = True
    if t.dim() != 2 or t.shape[1] != width:
        raise ValueError("%s must have %d columns, got shape %s"
                         % (what, width, tuple(t.shape)))
    if batch is not None and t.shape[0] != batch:
        raise ValueError("%s has %d rows, expected %d" % (what, t.shape[0], batch))
    return t, was_numpy


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def free_stream(device, tries=12, wait_s=0.25):
    """A stream whose work makes progress WHILE a resident kernel holds its own stream's hardware queue.  The runtime
    multiplexes streams onto a few hardware queues (round-robin by creation order, per priority): a producer stream that
    lands on the resident kernel's queue waits until the kernel leaves - by its watchdog, seconds later.  Which queue a new
    stream gets depends on how many streams the process created before, so asking for "another priority" is not enough
    (seen in the GPU suite: the same test passed or stalled with the number of streams earlier tests had made).  This
    tries candidate streams of both priorities with a one-word kernel and returns the first on which it completes within
    ``wait_s``; RuntimeError if none does."""
    import time
    torch = _torch()
    dev = device_of(device)
    probe = torch.zeros(1, dtype=torch.int32, device=dev)
    kept = []       # (stalled candidates stay alive until we return: a freed stream's queue slot would be handed out again)
    for k in range(tries):
        s = torch.cuda.Stream(device=dev, priority=-1 if k % 2 == 0 else 0)
        with torch.cuda.stream(s):
            probe.add_(1)
        t0 = time.time()
        while time.time() - t0 < wait_s:
            if s.query():
                return s
            time.sleep(0.002)
        kept.append(s)
    raise RuntimeError("no stream makes progress beside the resident kernel (%d candidates tried): is another kernel "
                       "occupying the device?" % tries)


class ResidentWatchdog(RuntimeError):
    """A resident run ended by its watchdog (``ticket[32] == 2``): some wave used up the poll budget waiting for a ticket."""


def resident_wait(run):
    """Wait for a resident run (the dict ``resident_start`` returned) to leave and say how: returns the number of ticks
    every wave finished; raises ``ResidentWatchdog`` when the kernel left by its watchdog instead (nobody published the
    next ticket within ``timeout_s`` - typically a feeder whose stream shares the kernel's hardware queue, see
    ``free_stream``).  ``ticket[32] == 1`` (the caller asked it to stop) is not an error."""
    run["stream"].synchronize()
    tk = run["ticket"].cpu()
    stop, ticks_done = int(tk[32]), int(tk[49])
    if stop == 2:
        raise ResidentWatchdog(
            "resident kernel left by its watchdog after %d tick(s): wave %d used up its budget of %d polls waiting for ticket "
            "%d (in_seq = %d).  Is the producer on a stream that makes progress beside the kernel (resident_feed_stream())?"
            % (ticks_done, int(tk[54]), (int(tk[50]) & 0xffffffff) | ((int(tk[51]) & 0xffffffff) << 32), ticks_done + 1,
               int(tk[0])))
    return ticks_done


class SingleSlot(object):
    """Persistent staging for the single-instance ``solve()`` call (B = 1): one pinned host
    buffer each way that the kernel reads and writes in place (pinned host memory is mapped into
    the device's address space on ROCm, same pointer), so a call is one launch and one stream
    synchronisation - no copy commands.  ``CLIK_SOLVE_STAGED=1`` restores the staged variant
    (pinned -> device copy, launch, device -> pinned copy)."""

    def __init__(self, device, n_in, n_out, n_int):
        import os
        torch = _torch()
        self.device = device
        self.zero_copy = os.environ.get("CLIK_SOLVE_STAGED", "0") != "1"
        self.h_in = torch.empty((max(n_in, 1),), dtype=torch.float64).pin_memory()
        self.d_in = torch.empty((max(n_in, 1),), dtype=torch.float64, device=device)
        # outputs: n_out doubles followed by n_int int32 (padded to 8 bytes)
        self.n_out, self.n_int = n_out, n_int
        nbytes = 8 * n_out + 8 * ((n_int + 1) // 2)
        self.d_out = torch.zeros((max(nbytes, 8),), dtype=torch.uint8, device=device)
        self.h_out = torch.zeros((max(nbytes, 8),), dtype=torch.uint8).pin_memory()
        self.in_np = self.h_in.numpy()
        out_np = self.h_out.numpy()
        self.out_f = out_np[:8 * n_out].view(np.float64)
        self.out_i = out_np[8 * n_out:8 * n_out + 4 * n_int].view(np.int32)
        # one device in the process: no device-guard context around the call (it costs ~3 us)
        import contextlib
        self._single_device = torch.cuda.device_count() == 1
        self._no_guard = contextlib.nullcontext()
        self._stream = None
        self.d_in_ptr = self.h_in.data_ptr() if self.zero_copy else self.d_in.data_ptr()
        self.d_out_ptr = self.h_out.data_ptr() if self.zero_copy else self.d_out.data_ptr()

    def in_ptr(self, offset_doubles):
        return C.c_void_p(self.d_in_ptr + 8 * offset_doubles)

    def out_ptr(self, offset_doubles):
        return C.c_void_p(self.d_out_ptr + 8 * offset_doubles)

    def int_ptr(self, index=0):
        return C.c_void_p(self.d_out_ptr + 8 * self.n_out + 4 * index)

    def guard(self):
        """Device context for the call (a no-op when the process has one device)."""
        return self._no_guard if self._single_device else _torch().cuda.device(self.device)

    def begin(self):
        """Stream of this call (torch's current stream, looked up once) as a launch argument."""
        self._stream = _torch().cuda.current_stream(self.device)
        if not self.zero_copy:
            self.d_in.copy_(self.h_in, non_blocking=True)
        return C.c_void_p(self._stream.cuda_stream)

    def upload(self):
        if not self.zero_copy:
            self.d_in.copy_(self.h_in, non_blocking=True)

    def download(self):
        torch = _torch()
        if not self.zero_copy:
            self.h_out.copy_(self.d_out, non_blocking=True)
        st = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
        self._stream = None
        st.synchronize()
