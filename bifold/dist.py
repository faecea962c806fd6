"""Data-parallel gradient synchronisation for the BiFold training step (new work: the
reference is single-process).

One process per GPU; parameters and optimiser state are replicated, every rank draws a
disjoint minibatch, and the only exchange is ONE sum of the trainable gradients per step.
Because all gradients live in a flat arena (models/arena.py), a bucket is just a slice:
no flatten / unflatten copies.  Buckets follow the backward order of the fused tail --
decoders, then fusion layers depth-1 ... 0, then the input-side parameters -- and each is
all-reduced (RCCL over xGMI; `nccl` backend == RCCL on ROCm) on a side HIP stream as soon
as the backward pass reports it complete, while the remaining layers are still computing.
The optimiser waits on the last bucket only.

Each rank's loss is a mean over its local batch, so the AVERAGE of the per-rank gradients
equals the single-process gradient of the concatenated batch.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_process_group_from_env(device_backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world).  Reads the torchrun environment; a bare `python bench.py`
    is world 1 and needs no process group."""
    import bifold_amd  # noqa: F401  (sets GPU_MAX_HW_QUEUES -- the one place that does -- before the first HIP call)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("BIFOLD_FORCE_PG") == "1"      # rehearse the RCCL path on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = device_backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world > 1 or force:
        reserve_cus_for_collectives(local_rank)
    return rank, local_rank, world


def reserve_cus_for_collectives(local_rank: int = 0) -> Optional[int]:
    """`BIFOLD_P8_RESERVE_CUS=k`: with a process group alive, the persistent GEMMs run one workgroup on each of (CUs - k) CUs
    instead of all of them, so that RCCL's reduction kernels find k CUs that no resident GEMM holds (a persistent GEMM
    takes a whole CU: 225 registers x 2 waves per SIMD and 128 KB of LDS; without the knob the collective's
    workgroups are time-sliced into GEMM boundaries like any other kernel).  Implemented through the library's own grid cap
    (`BIFOLD_P8_GRID`, read once at first use): must run before the first GEMM.  Returns the cap it set, or None."""
    k = int(os.environ.get("BIFOLD_P8_RESERVE_CUS", "0") or 0)
    if k <= 0 or "BIFOLD_P8_GRID" in os.environ or not torch.cuda.is_available():
        return None
    n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
    cap = max(8, n_cu - k)
    os.environ["BIFOLD_P8_GRID"] = str(cap)
    return cap


def layer_buckets(names: List[str], offsets, depth: int, count: int) -> List[Tuple[int, int]]:
    """Arena slices in the order the fused backward completes them.
    Index 0: everything after the last fusion layer (the decoders); 1..depth: fusion layers
    depth-1 ... 0; depth+1: everything before layer 0 (tokens, positional embeddings,
    project, type embeddings).  `offsets[name] = (start, numel)`; slices cover the
    256-byte-aligned padding too, so together they tile [0, count)."""
    marker = "transformer_encoder.layers."
    starts = {}
    for n in names:
        if marker in n:
            i = int(n.split(marker)[1].split(".")[0])
            starts[i] = min(starts.get(i, 1 << 62), offsets[n][0])
    if sorted(starts) != list(range(depth)):
        raise ValueError("parameter names do not contain the expected fusion layers")
    # start of the first parameter after the last layer
    after = min((offsets[n][0] for n in names if marker not in n and offsets[n][0] > starts[depth - 1]),
                default=count)
    # bucket 0 leaves as soon as the LAST fusion layer reports: it may only hold parameters whose gradients
    # the backward has written by then, i.e. the decoders.  Anything else registered behind the fusion layers
    # (a new head, an arena built in another order) goes with the final bucket instead of being averaged stale.
    # (the mask head of compute_mask=True is a decoder in this sense: its gradient is written with the heads')
    is_dec = lambda n: "_decoder." in n or "mask_head." in n   # noqa: E731
    late = [n for n in names if marker not in n and offsets[n][0] >= after and not is_dec(n)]
    if late:
        after_dec = min(offsets[n][0] for n in late)
        early = [n for n in names if offsets[n][0] >= after_dec and is_dec(n)]
        if early:
            raise ValueError(f"decoder parameters {early[:2]}... are interleaved with {late[:2]}... behind the "
                             f"fusion layers: cannot bucket the gradient arena in backward order")
        buckets = [(after, after_dec)]
        tail_extra = (after_dec, count)
    else:
        buckets = [(after, count)]
        tail_extra = None
    for i in reversed(range(depth)):
        end = starts[i + 1] if i + 1 < depth else after
        buckets.append((starts[i], end))
    buckets.append((0, starts[0]))
    if tail_extra is not None:
        buckets.append(tail_extra)
    return buckets


def gradient_sync_for(model, group: Optional[dist.ProcessGroup] = None) -> "GradSync":
    """GradSync over `model`'s gradient arena, bucketed per fusion layer.  The layout is taken
    from the model's own parameter names (matched to arena slots by identity), so it does not
    matter whether the model or the optimiser created the arena first."""
    arena = model.trainable_arena()
    named = model.trainable_params()
    names = [n for n, _ in named]
    offsets = {n: arena.offsets[arena.name_of(p)] for n, p in named}
    depth = model.pick_place.fusion.depth
    sync = GradSync(arena.grad, layer_buckets(names, offsets, depth, arena.count), depth, group)
    model.register_grad_ready_hook(sync.on_ready)
    # fp16 mode: the overflow flag that guards the optimiser step (models/gradscale.py) must read the same on every
    # rank, or one replica drops a step the others apply -- it travels (MAX) behind the last gradient bucket
    from .models.gradscale import scaler_of
    scaler = scaler_of(model)
    if scaler is not None:
        sync.flag_getter = lambda: scaler.nonfinite
    return sync


class GradSync:
    """Overlapped all-reduce of a flat gradient arena.

    usage:  sync = GradSync(arena.grad, buckets);  model.register_grad_ready_hook(sync.on_ready)
            ... loss.backward() ...;  sync.finish()  (before optimizer.step())
    `on_ready(i)`: i = depth-1 ... 0 as fusion layer i completes, then -1 at the end.

    `buckets` (layer_buckets) are adjacent arena slices in backward order; `coalesce` consecutive
    layers travel in ONE all-reduce (default 4, env BIFOLD_DDP_COALESCE): 105 MB of gradients take
    ~0.6 ms on xGMI, so two or three large collectives hide as well under the remaining backward as
    ten small ones with a third of the launches, stream waits and CU grabs (with the RCCL path forced at
    world 1 the step costs the same for 10, 3 or 1 collectives; what did cost 4.6 % was stream-to-queue
    aliasing, fixed by GPU_MAX_HW_QUEUES=8 -- DESIGN.md §7).
    """

    def __init__(self, flat_grad: torch.Tensor, buckets: List[Tuple[int, int]], depth: int,
                 group: Optional[dist.ProcessGroup] = None, coalesce: Optional[int] = None):
        self.grad = flat_grad
        self.buckets = buckets
        self.depth = depth
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.on_gpu = flat_grad.is_cuda
        self.avg_native = (not dist.is_initialized()) or dist.get_backend(group) == "nccl"    # ReduceOp.AVG is RCCL's
        self.side = torch.cuda.Stream(device=flat_grad.device) if self.on_gpu else None
        self.pending: List = []
        self.flag_getter = None          # () -> int32 [1] tensor MAX-reduced with the final bucket, or None
        # read-out for bench.py (`measure = True`): collectives and bytes launched, and the time the COMPUTE stream spends
        # waiting in finish() -- an event pair around the waits, read after the run (no sync inside the step)
        self.measure = False
        self.stats = {"launches": 0, "bytes": 0, "steps": 0}
        self._wait_events: List = []
        if coalesce is None:
            coalesce = int(os.environ.get("BIFOLD_DDP_COALESCE", "4"))
        self.coalesce = max(1, coalesce)
        # plan: trigger layer -> merged slice.  Entry j of `buckets` (j = 0: decoders, j = 1 + k: layer
        # depth-1-k) is complete once layer depth-1-k (for j = 0: layer depth-1) has reported; a group of
        # entries is sent when its LAST member is complete.  What is left goes with the final call (-1).
        self.plan = {}
        entries = list(range(1 + depth))                 # all but the input-side bucket
        groups, cur = [], []
        for j in entries:
            cur.append(j)
            layers_in = sum(1 for x in cur if x >= 1)
            if layers_in >= self.coalesce:
                groups.append(cur)
                cur = []
        tail = cur + list(range(1 + depth, len(buckets)))  # remainder + input-side (+ late) parameters
        for g in groups:
            last_layer = depth - 1 - (g[-1] - 1)
            self.plan[last_layer] = self._merge([buckets[j] for j in g])
        self.final = self._merge([buckets[j] for j in tail]) if depth > 0 or len(buckets) == 1 else None
        if depth == 0:                                    # no fusion layers: everything at the end
            self.final = self._merge(list(buckets))

    @staticmethod
    def _merge(spans: List[Tuple[int, int]]) -> List[Tuple[int, int]]:
        """Adjacent slices -> as few contiguous ranges as possible."""
        out: List[Tuple[int, int]] = []
        for a, b in sorted(s for s in spans if s[1] > s[0]):
            if out and out[-1][1] == a:
                out[-1] = (out[-1][0], b)
            else:
                out.append((a, b))
        return out

    def _launch(self, spans: List[Tuple[int, int]]) -> None:
        if self.world == 1 and os.environ.get("BIFOLD_FORCE_PG") != "1":
            return
        if os.environ.get("BIFOLD_DDP_DRYRUN") == "1":      # diagnosis: process group alive, no collective issued
            if self.world > 1:                              # replicas would silently train on un-averaged gradients
                raise RuntimeError("BIFOLD_DDP_DRYRUN=1 is a single-rank rehearsal knob (BIFOLD_FORCE_PG=1); "
                                   f"refusing to skip the gradient all-reduce with world size {self.world}")
            return
        for a, b in spans:
            view = self.grad[a:b]
            self.stats["launches"] += 1
            self.stats["bytes"] += (b - a) * view.element_size()
            if self.on_gpu and self.avg_native:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream())
                self.side.wait_event(ready)
                with torch.cuda.stream(self.side):
                    work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                self.pending.append(work)
            elif self.on_gpu:            # a backend without AVG on device tensors (gloo: a rehearsal of N ranks on ONE GPU)
                torch.cuda.current_stream().synchronize()
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                view.div_(self.world)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                view.div_(self.world)

    def _launch_flag(self) -> None:
        flag = self.flag_getter() if self.flag_getter is not None else None
        if flag is None or (self.world == 1 and os.environ.get("BIFOLD_FORCE_PG") != "1") or \
                os.environ.get("BIFOLD_DDP_DRYRUN") == "1":
            return
        if self.on_gpu and self.avg_native:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                work = dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
            self.pending.append(work)
        else:
            if self.on_gpu:
                torch.cuda.current_stream().synchronize()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)

    def on_ready(self, layer: int) -> None:
        if layer >= 0:
            spans = self.plan.get(layer)
            if spans:
                self._launch(spans)
        else:
            if self.final:
                self._launch(self.final)
            self._launch_flag()           # after the last un-scaling of this backward: the count is complete

    def finish(self) -> None:
        """Make the compute stream wait for every outstanding bucket."""
        timed = self.measure and self.on_gpu and bool(self.pending)
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        for w in self.pending:
            w.wait()
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            self._wait_events.append((e0, e1))
        self.stats["steps"] += 1
        self.pending.clear()

    def describe_plan(self) -> dict:
        """What the first N > 1 record needs to explain itself: the collectives of one step in launch order (trigger =
        the fusion layer whose completed backward releases the slice, -1 = the end of the backward), their sizes, the
        backend and the RCCL version the process group initialised."""
        esz = self.grad.element_size()
        order = sorted((k for k in self.plan), reverse=True)           # the backward walks the layers downwards
        launches = [{"after_layer": int(k), "bytes": [int((b - a) * esz) for a, b in self.plan[k]]} for k in order]
        if self.final:
            launches.append({"after_layer": -1, "bytes": [int((b - a) * esz) for a, b in self.final]})
        backend, version = None, None
        if dist.is_initialized():
            backend = dist.get_backend(self.group)
            if backend == "nccl":
                try:
                    version = ".".join(str(x) for x in torch.cuda.nccl.version())
                except Exception:                                          # a build without the binding: say so, do not fail
                    version = "unknown"
        return {"launches": launches, "coalesce_layers": self.coalesce, "backend": backend, "rccl_version": version,
                "world": self.world, "overflow_flag": "int32 MAX-reduced behind the last bucket" if self.flag_getter else None}

    def reset_stats(self) -> None:
        self.stats = {"launches": 0, "bytes": 0, "steps": 0}
        self._wait_events.clear()

    def comm_report(self) -> dict:
        """After a synchronize: what the run launched per step and how long the compute stream stood in finish() per step
        (the part of the communication that was NOT hidden under the backward)."""
        steps = max(1, self.stats["steps"])
        exposed = sum(a.elapsed_time(b) for a, b in self._wait_events) if self._wait_events else None
        return {"exposed_ms_per_step": None if exposed is None else round(exposed / steps, 4),
                "buckets": round(self.stats["launches"] / steps, 2), "bytes": int(self.stats["bytes"] / steps),
                "steps": steps, "plan": self.describe_plan(),
                "note": "exposed = time the compute stream waits in GradSync.finish() for the side stream's collectives "
                        "(HIP events around the waits); buckets / bytes = all-reduce launches and gradient bytes per step"}
