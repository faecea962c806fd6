"""Data-parallel gradient averaging with world_size 2 over gloo (CPU): the bucket schedule of
GradSync must leave every rank with the mean of the per-rank gradient arenas, which is the
single-process gradient of the concatenated batch when each rank's loss is a local mean."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bifold import dist as bdist
    r, lr, w = bdist.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    depth = 3
    names, offsets, off = [], {}, 0
    for n, cnt in [("text_token", 64), ("project.weight", 300)] + \
                  [(f"pick_place.fusion.transformer_encoder.layers.{i}.{j}.w", 200 + i) for i in range(depth) for j in (0, 1)] + \
                  [("pick_place.pick_decoder.decoder_net.0.weight", 130)]:
        names.append(n); offsets[n] = (off, cnt); off += (cnt + 63) // 64 * 64
    grad = torch.full((off,), float(rank + 1)) + torch.arange(off) * (rank + 1) * 1e-3
    expect = (torch.full((off,), 1.5) + torch.arange(off) * 1.5e-3)
    ok = True
    for coalesce in (1, 2, 3, None):              # one all-reduce per layer ... everything in the final one
        g = grad.clone()
        s2 = bdist.GradSync(g, bdist.layer_buckets(names, offsets, depth, off), depth, coalesce=coalesce)
        sent = sorted(sum(list(s2.plan.values()), []) + s2.final)
        ok = ok and sent[0][0] == 0 and sent[-1][1] == off and all(sent[i][1] == sent[i + 1][0]
                                                                  for i in range(len(sent) - 1))
        for layer in reversed(range(depth)):
            s2.on_ready(layer)
        s2.on_ready(-1)
        s2.finish()
        ok = ok and torch.allclose(g, expect, rtol=1e-6)
    sync = bdist.GradSync(grad, bdist.layer_buckets(names, offsets, depth, off), depth)
    for layer in reversed(range(depth)):          # the order the fused backward reports completion
        sync.on_ready(layer)
    sync.on_ready(-1)
    sync.finish()
    ok = ok and torch.allclose(grad, expect, rtol=1e-6)
    # a second step must work identically (state reset by finish())
    grad.copy_(torch.full((off,), float(10 * (rank + 1))))
    for layer in reversed(range(depth)):
        sync.on_ready(layer)
    sync.on_ready(-1)
    sync.finish()
    ok = ok and torch.allclose(grad, torch.full((off,), 15.0))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


class _HostArena:
    """The memory layout of models/arena.ParamArena on the host (the real one lives in HBM and casts its bf16
    shadow with a HIP kernel): enough for dist.gradient_sync_for -- `.grad`, `.offsets`, `.name_of`, `.count`."""

    def __init__(self, named):
        self.offsets, self._by_id, off = {}, {}, 0
        for n, p in named:
            self.offsets[n] = (off, p.numel())
            self._by_id[id(p)] = n
            off += (p.numel() + 63) // 64 * 64
        self.count = off
        self.grad = torch.zeros(off)

    def name_of(self, p):
        return self._by_id[id(p)]


def _hook_worker(rank, world, port, out):
    """A REAL RGBOnly (tiny, on the host) whose `register_grad_ready_hook` / `_grad_ready` drive GradSync exactly as
    its fused backward does on the GPU: layers depth-1 ... 0, then -1.  Each layer's gradients are written just
    before its report, so a bucket that left too early would be averaged stale and the check below would fail."""
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bifold import dist as bdist
    from bifold.models import Models
    from bifold.models.clip_encoder import CLIP_CONFIGS
    bdist.init_process_group_from_env("gloo")
    CLIP_CONFIGS["tiny-ddp"] = dict(embed_dim=32, image_resolution=64, vision_layers=1, vision_width=64,
                                    vision_patch_size=16, context_length=77, vocab_size=64, transformer_width=64,
                                    transformer_heads=1, transformer_layers=1)
    cfg = dict(name="rgb_clip", image_size=64, is_bimanual=True, requires_graph=False, text_encoder="tiny-ddp",
               patch_size=16, depth=5, heads=2, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="bf16")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = Models.get_by_name(cfg, device=torch.device("cpu"))
    arena = _HostArena(model.trainable_params())
    model.trainable_arena = lambda: arena
    ok = True
    for coalesce in ("1", "2", "4", "8"):
        os.environ["BIFOLD_DDP_COALESCE"] = coalesce
        model._grad_hooks.clear()
        sync = bdist.gradient_sync_for(model)
        assert model._grad_hooks == [sync.on_ready]
        depth = model.pick_place.fusion.depth
        for step in range(2):
            arena.grad.fill_(float("nan"))             # anything averaged before it was written stays NaN
            value = lambda n: (rank + 1) * (1 + step) * (1 + (hash(n) % 97) / 97.0)
            def write(pred):
                for n, p in model.trainable_params():
                    if pred(n):
                        o, cnt = arena.offsets[n]
                        arena.grad[o:o + (cnt + 63) // 64 * 64] = value(n)
            write(lambda n: "_decoder." in n)          # the backward starts with the decoders ...
            for i in reversed(range(depth)):           # ... then fusion layers depth-1 ... 0 ...
                write(lambda n, i=i: f"layers.{i}." in n)
                model._grad_ready(i)
            write(lambda n: "_decoder." not in n and "layers." not in n)   # ... then the input side
            model._grad_ready(-1)
            sync.finish()
            for n, p in model.trainable_params():
                o, cnt = arena.offsets[n]
                want = 1.5 * (1 + step) * (1 + (hash(n) % 97) / 97.0)
                ok = ok and bool(torch.allclose(arena.grad[o:o + cnt], torch.full((cnt,), want), rtol=1e-6))
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_model_hooks_drive_gradsync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    os.environ["PYTHONHASHSEED"] = "0"               # hash(name) must agree across the two ranks
    mp.spawn(_hook_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _flag_worker(rank, world, port, out):
    """fp16 mode under data parallelism: the overflow flag that guards the optimiser step is MAX-reduced behind the
    last gradient bucket, so a backward that overflowed on ONE rank drops the step on EVERY rank (and a clean step
    stays clean).  Real model hooks, host arena, gloo."""
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bifold import dist as bdist
    from bifold.models import Models
    from bifold.models.clip_encoder import CLIP_CONFIGS
    from bifold.models.gradscale import scaler_of
    bdist.init_process_group_from_env("gloo")
    CLIP_CONFIGS["tiny-ddp"] = dict(embed_dim=32, image_resolution=64, vision_layers=1, vision_width=64,
                                    vision_patch_size=16, context_length=77, vocab_size=64, transformer_width=64,
                                    transformer_heads=1, transformer_layers=1)
    cfg = dict(name="rgb_clip", image_size=64, is_bimanual=False, requires_graph=False, text_encoder="tiny-ddp",
               patch_size=16, depth=3, heads=2, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="fp16")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = Models.get_by_name(cfg, device=torch.device("cpu"))
    arena = _HostArena(model.trainable_params())
    model.trainable_arena = lambda: arena
    sync = bdist.gradient_sync_for(model)
    scaler = scaler_of(model)
    assert scaler is not None and sync.flag_getter is not None
    scaler._plan()                                     # what the first backward does: the flag exists
    assert arena.grad_scaler is scaler
    depth = model.pick_place.fusion.depth
    seen = []
    for step, bad_rank in enumerate((1, None, 0)):
        arena.grad.fill_(float(rank + 1))
        if bad_rank == rank:
            scaler.nonfinite[0] = 7 + rank                 # this rank's un-scaling counted non-finite results
        for i in reversed(range(depth)):
            model._grad_ready(i)
        model._grad_ready(-1)
        sync.finish()
        seen.append(int(scaler.nonfinite.item()))
        scaler.nonfinite.zero_()                           # bf_gradscale_update's last act
    out[rank] = seen == [8, 0, 7] and bool(torch.allclose(arena.grad, torch.full_like(arena.grad, 1.5)))
    dist.barrier()
    dist.destroy_process_group()


def test_overflow_flag_reaches_every_rank_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flag_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


class _HostModel(torch.nn.Module):
    """A host stand-in with the data-parallel contract of the real models (bifold/models/__init__.py: `trainable_params`,
    `trainable_arena`, `register_grad_ready_hook` / `_grad_ready`): a stack of `depth` residual MLP layers named like the
    fusion transformer's, a decoder behind them and a projection in front; parameters AND gradients are views of flat
    arenas, and the backward reports layers depth-1 ... 0, then -1, from autograd's post-accumulate hooks -- the order
    the fused backward reports them in."""

    def __init__(self, depth=3, width=16):
        super().__init__()
        g = torch.Generator().manual_seed(3407)
        shapes = [("project.weight", (width, 8))]
        for i in range(depth):
            shapes += [(f"pick_place.fusion.transformer_encoder.layers.{i}.0.w", (width, width)),
                       (f"pick_place.fusion.transformer_encoder.layers.{i}.1.w", (width, width))]
        shapes += [("pick_place.pick_decoder.decoder_net.0.weight", (1, width))]
        self.depth = depth
        self.offsets, off = {}, 0
        for n, shp in shapes:
            cnt = int(torch.tensor(shp).prod())
            self.offsets[n] = (off, cnt)
            off += (cnt + 63) // 64 * 64
        self.count = off
        self.flat = torch.zeros(off)
        self.grad = torch.zeros(off)
        self._named, self._by_id = [], {}
        for n, shp in shapes:
            o, cnt = self.offsets[n]
            p = torch.nn.Parameter(self.flat[o:o + cnt].view(shp))
            with torch.no_grad():
                p.copy_(torch.randn(shp, generator=g) * 0.3)
            p.grad = self.grad[o:o + cnt].view(shp)             # .grad is a view of the arena, as in models/arena.py
            self._named.append((n, p))
            self._by_id[id(p)] = n
            self.register_parameter(n.replace(".", "_"), p)
        self._grad_hooks = []
        self._left = {}
        for n, p in self._named:
            p.register_post_accumulate_grad_hook(lambda q, n=n: self._param_done(n))
        import types
        self.pick_place = types.SimpleNamespace(fusion=types.SimpleNamespace(depth=depth))   # (what gradient_sync_for reads)

    # --- the contract dist.gradient_sync_for uses
    def trainable_params(self):
        return list(self._named)

    def trainable_arena(self):
        return self

    def name_of(self, p):
        return self._by_id[id(p)]

    def register_grad_ready_hook(self, fn):
        self._grad_hooks.append(fn)

    def _grad_ready(self, layer):
        for fn in self._grad_hooks:
            fn(layer)

    def _param_done(self, name):
        marker = "transformer_encoder.layers."
        if marker in name:
            i = int(name.split(marker)[1].split(".")[0])
            self._left[i] = self._left.get(i, 2) - 1
            if self._left[i] == 0:
                self._left[i] = 2
                self._grad_ready(i)
        elif name == "project.weight":                          # the input side is the last thing the backward reaches
            self._grad_ready(-1)

    def forward(self, sample):
        p = dict(self._named)
        x = sample["x"] @ p["project.weight"].t()
        for i in range(self.depth):
            pre = f"pick_place.fusion.transformer_encoder.layers.{i}."
            x = x + torch.tanh(x @ p[pre + "0.w"].t()) @ p[pre + "1.w"].t()
        return {"pick_heatmap": torch.sigmoid(x @ p["pick_place.pick_decoder.decoder_net.0.weight"].t())}


def _bench_loop_worker(rank, world, port, out):
    """bench.py's OWN step (bench.train_step: forward, loss, zero_grad, backward with the bucket hooks, finish, optimiser
    step) under world-2 gloo: each rank sees different data, so only the averaged gradients can keep the replicas equal."""
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from bifold import dist as bdist
    bdist.init_process_group_from_env("gloo")
    model = _HostModel()
    start = model.flat.clone()
    sync = bdist.gradient_sync_for(model)
    sync.measure = True
    opt = torch.optim.Adam([p for _, p in model.trainable_params()], lr=1e-2)
    # (torch's Adam with set_to_none=False keeps .grad the arena view; zero_grad() then zeroes the arena in place)
    opt_zero = opt.zero_grad
    opt.zero_grad = lambda: opt_zero(set_to_none=False)
    g = torch.Generator().manual_seed(100 + rank)               # per-rank data, as bench.py seeds its batches
    loss_fn = lambda o, s: (torch.nn.functional.binary_cross_entropy(o["pick_heatmap"], s["t"]), {})
    losses = []
    for _ in range(3):
        sample = {"x": torch.randn(32, 8, generator=g), "t": torch.rand(32, 1, generator=g)}
        losses.append(float(bench.train_step(model, loss_fn, opt, sync, sample)))
    rep = sync.comm_report()
    gathered = [torch.zeros_like(model.flat) for _ in range(world)]
    dist.all_gather(gathered, model.flat)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    moved = float((model.flat - start).abs().max()) > 1e-3
    lg = [None] * world
    dist.all_gather_object(lg, losses)
    differ = lg[0] != lg[1]                                     # the ranks really saw different batches
    out[rank] = bool(same and moved and differ and rep["steps"] == 3 and rep["buckets"] >= 1 and rep["bytes"] > 0)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_loop_keeps_replicas_identical_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bench_loop_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment must start its two ranks itself (the driver's
    command).  Without a GPU each rank stops at "needs an MI355X" -- AFTER it has joined the 2-rank rendezvous the
    parent set up -- and the parent relays the failure as its exit status."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout, r.stderr[-3000:]
    else:
        assert r.returncode != 0
        assert "disagrees" not in r.stderr and "launch with" not in r.stderr
        assert r.stderr.count("needs an MI355X") >= 2 or "device" in r.stderr.lower(), r.stderr[-3000:]
