"""bifold_amd -- MI355X-native (gfx950) implementation of BiFold's VLM training step.

Mirrors the reference's plugin surface (`bifold.models`, `bifold.losses`, `bifold.optim`,
`python -m bifold`) on top of libbifold_hip.so; see DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"
