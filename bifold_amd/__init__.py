"""bifold_amd -- the native layer of the MI355X (gfx950) build of BiFold's VLM training step:
`csrc/` (hand-written HIP kernels + the C ABI of include/bifold_hip.h), `lib/libbifold_hip.so`
and `hip/` (the ctypes binding and the tensor-level wrappers).

The plugin surface the reference trainer imports lives in the `bifold` package next to this one
(`bifold.models`, `bifold.losses`, `bifold.optim`, `python -m bifold`).
`bifold_amd.models`, `.losses`, `.optim`, `.dist`, `.config`, `.targets` remain importable as aliases of
the `bifold.*` modules of the same name (round-1 spelling).
"""
import importlib as _importlib
import importlib.abc as _abc
import importlib.util as _util
import os as _os
import sys as _sys

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue do
# not overlap.  The step runs the text tower beside the vision tower on a second stream; with a live RCCL
# communicator and the gradient side stream there are more streams than queues, the two towers land on one
# queue and serialise (measured: -4.6 % of the step at world size 1, -0.8 % with 8 queues).  The runtime
# reads the variable when it initialises, i.e. at the first HIP call, which comes after this import.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.2.0"

_ALIASED = ("models", "losses", "optim", "dist", "config", "targets")


class _AliasLoader(_abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        module = _importlib.import_module(self.target)     # the bifold.* module itself, not a copy
        self.real_spec = module.__spec__
        return module

    def exec_module(self, module):
        module.__spec__ = self.real_spec                   # the import system just pointed it at the alias


class _AliasFinder(_abc.MetaPathFinder):
    """`bifold_amd.<models|losses|optim|...>[.sub]` -> the module object of `bifold.<...>[.sub]`."""

    def find_spec(self, fullname, path=None, target=None):
        parts = fullname.split(".")
        if len(parts) < 2 or parts[0] != "bifold_amd" or parts[1] not in _ALIASED:
            return None
        real = "bifold." + ".".join(parts[1:])
        if _util.find_spec(real) is None:
            return None
        return _util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in _sys.meta_path):
    _sys.meta_path.insert(0, _AliasFinder())
