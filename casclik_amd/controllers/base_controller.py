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
    return torch.device(device)


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
