"""The side HIP streams of the training step.

HIP maps streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, default 4), and a process that keeps more
queues busy than the hardware scheduler can map at once is time-sliced with wave save / restore: with a live RCCL
communicator and GPU_MAX_HW_QUEUES=8 every kernel of the step ran 1.3-5x longer (45 ms per step instead of 37; DESIGN.md
section 7).  The step therefore uses as few streams as its concurrency needs:

  main                      the dependent chain (vision tower, fusion, backward dX chain, optimiser)
  side (this module)        work that runs BESIDE the chain and never beside itself: the text tower and the decoder
                            collapse in the forward, the decoder-collapse backward and the weight-gradient GEMMs in the
                            backward
  comm (bifold/dist.py)     the gradient all-reduces (RCCL adds its own internal stream)

(One stream per role measured no better and costs hardware queues: removed in round 6.)

BIFOLD_SERIAL_STREAMS=1 (or `set_serial(True)`) folds the side work into the main stream: the PROFILING mode in which a
kernel's duration is its own cost instead of its share of a contended machine (bench.py's roofline sample,
tools/serial_stats.sh).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

_SERIAL = os.environ.get("BIFOLD_SERIAL_STREAMS", "0") == "1"
_streams: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}


def side_stream(device, role: str) -> "torch.cuda.Stream":
    """The stream for `role` ("text", "decoder", "dw") on `device`: ONE shared side stream."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, "side")
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=torch.device("cuda", index))
    return _streams[key]


def serial() -> bool:
    """True: towers, decoder collapse and weight-gradient GEMMs all run on the main stream (profiling mode)."""
    return _SERIAL


def set_serial(on: bool) -> bool:
    """Switch the profiling mode; returns the previous setting."""
    global _SERIAL
    old, _SERIAL = _SERIAL, bool(on)
    return old
