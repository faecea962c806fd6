"""bifold_amd -- MI355X-native (gfx950) implementation of BiFold's VLM training step.

Mirrors the reference's plugin surface (`bifold.models`, `bifold.losses`, `bifold.optim`,
`python -m bifold`) on top of libbifold_hip.so; see DESIGN.md and INTEGRATION.md.
"""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue do
# not overlap.  The step runs the text tower beside the vision tower on a second stream; with a live RCCL
# communicator and the gradient side stream there are more streams than queues, the two towers land on one
# queue and serialise (measured: -4.6 % of the step at world size 1, -0.8 % with 8 queues).  The runtime
# reads the variable when it initialises, i.e. at the first HIP call, which comes after this import.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
