#!/usr/bin/env python3
"""Runs tools/probe_profiler_stretch.hip (built as tools/_build/libprobe_profiler_stretch.so) inside a Python process:
    python3 tools/profiler_stretch.py
    rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/profiler_stretch.py
See the .hip file for what is measured (the profiler's stretch of a us-sized kernel)."""
import ctypes
import os

import torch        # (the HIP runtime the process uses, as every other profiled command of this repo)

torch.cuda.init()
torch.zeros(1, device="cuda")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libprobe_profiler_stretch.so"))
raise SystemExit(lib.probe_profiler_stretch_run())
