"""The Trainer of `python -m bifold` on the MI355X: gradient clipping through the
UNCHANGED torch call, LinearWarmup inside the loop, periodic evaluation (`get_action` -> pixel metrics ->
best.pth), checkpoints in the reference's schema, auto-resume from the Hydra-style run directory, and the
data-parallel hooks with a live RCCL group (world size 1, BIFOLD_FORCE_PG=1)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY_CLIP = dict(embed_dim=32, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
                 context_length=77, vocab_size=49408, transformer_width=64, transformer_heads=1, transformer_layers=2)
OVERRIDES = ["model=rgb_clip", "dataset@train_dataset=synthetic", "model.text_encoder=tiny-trainer",
             "model.image_size=64", "train_dataset.image_size=64", "train_dataset.length=64", "model.depth=2",
             "model.heads=2", "batch_size=8", "test_batch_size=8", "epochs=3", "gradient_clip=0.05",
             "scheduler=linear_warmup", "save_epochs=1", "eval_epochs=1", "simulator=null", "use_wandb=false",
             "model.dtype=fp32", "ddp=false"]


@pytest.fixture()
def run_home(tmp_path, monkeypatch):
    from bifold.models.clip_encoder import CLIP_CONFIGS
    CLIP_CONFIGS["tiny-trainer"] = TINY_CLIP
    monkeypatch.setenv("HOME", str(tmp_path))
    cwd = os.getcwd()
    yield tmp_path
    os.chdir(cwd)


def _trainer(overrides):
    import bifold.__main__ as M
    from bifold.config import compose
    cfg, hydra = compose(M.CONF_DIR, overrides, return_hydra=True)
    os.makedirs(hydra.run.dir, exist_ok=True)
    os.chdir(hydra.run.dir)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return M.Trainer(cfg, overrides), hydra


def test_trainer_clip_warmup_eval_save_and_resume(gpu, run_home, monkeypatch):
    import bifold.__main__ as M
    import bifold.optim as BO
    from bifold_amd.hip import ops
    calls = {"fused": 0}
    real = ops.clip_grad_norm_

    def counting(flat, max_norm):
        calls["fused"] += 1
        return real(flat, max_norm)
    monkeypatch.setattr(ops, "clip_grad_norm_", counting)

    # ---- an uninterrupted 3-epoch run through main(): clip + warm-up + eval each epoch + checkpoints
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = M.main(list(OVERRIDES))
    run_dir = os.getcwd()
    assert run_dir.startswith(str(run_home)) and "model=rgb_clip" in os.path.basename(run_dir)
    assert "seed" not in os.path.basename(run_dir) and "use_wandb" not in os.path.basename(run_dir)
    assert os.path.isfile("config.yaml") and os.path.isfile("checkpoints/last.pth") and os.path.isfile("checkpoints/best.pth")
    assert os.path.isfile("eval_synthetic.yaml")
    steps = 3 * len(t.train_dataloader)
    assert calls["fused"] == steps          # torch.nn.utils.clip_grad_norm_(model.parameters(), c) took the arena path
    assert torch.nn.utils.clip_grad_norm_ is BO.clip_grad_norm_
    ck = torch.load("checkpoints/last.pth", map_location="cpu", weights_only=False)
    # the reference's schema (bifold/__main__.py:304-325)
    assert set(ck) >= {"epoch", "random_states", "model", "optimizer", "scheduler", "best_eval"}
    assert ck["epoch"] == 3 and len(ck["random_states"]) == 4
    assert isinstance(ck["random_states"][2], torch.Tensor) and isinstance(ck["random_states"][3], torch.Tensor)
    assert ck["scheduler"]["last_step"] == t.scheduler.last_step and "optimizer" not in ck["scheduler"]
    assert ck["optimizer"]["state"][0]["step"].item() == steps
    final = {k: v.clone() for k, v in t.model.state_dict().items()}
    assert t.metrics.best_eval is not None and np.isfinite(t.metrics.best_eval)

    # ---- the same command again: auto-resume finds the finished run, trains nothing, evaluates
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t2 = M.main(list(OVERRIDES))
    assert os.getcwd() == run_dir and t2.start_epoch == 3
    for k, v in final.items():
        assert torch.equal(t2.model.state_dict()[k].cpu(), v.cpu()), k

    # ---- an interrupted run: epoch 0, save, new process state, resume -> same parameters as 2 straight epochs
    ov = [o for o in OVERRIDES if not o.startswith("epochs=")] + ["epochs=2", "eval_epochs=null"]
    ta, _ = _trainer(ov)
    ta.prepare_train()
    ta.train()
    straight = {k: v.clone() for k, v in ta.model.state_dict().items() if "clip_encoder" not in k}
    os.remove("checkpoints/last.pth")
    tb, _ = _trainer(ov)
    tb.prepare_train()
    tb.train_epoch(0)
    tb.save_model(0)
    tc, _ = _trainer(ov)
    tc.prepare_train()
    assert tc.start_epoch == 1 and tc.optimizer.step_count == len(tc.train_dataloader)
    assert tc.scheduler.last_step == tb.scheduler.last_step
    tc.train()
    for k, v in straight.items():
        got = tc.model.state_dict()[k]
        assert (got - v).abs().max().item() <= 1e-6 * (v.abs().max().item() + 1e-12) + 1e-9, k


def test_trainer_default_fp16_mode_trains_and_survives_overflow(gpu, run_home, capsys):
    """`python -m bifold` with the shipped `model.dtype: fp16`: the loss-scaled backward trains (mean loss falls over
    the epochs, no step dropped); with a loss scale far beyond binary16's range every backward overflows and every
    step is DROPPED on the device -- parameters and moments stay bit for bit what they were, the scale halves per
    step, the Trainer reports the drops at the end of the epoch and does not raise (round 3 applied the non-finite
    gradients and raised an epoch later)."""
    import bifold.__main__ as M
    import re
    import warnings
    ov = [o for o in OVERRIDES if not o.startswith(("model.dtype=", "epochs=", "gradient_clip="))] + ["epochs=4"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = M.main(list(ov))
    assert t.model.compute_dtype == torch.float16 and t.model._grad_scaler.read()["skipped"] == 0
    means = [float(x) for x in re.findall(r"mean loss ([0-9.]+)", capsys.readouterr().out)]
    assert len(means) == 4 and means[-1] < means[0], means
    os.chdir(str(run_home))
    bad = [o for o in ov if not o.startswith(("epochs=", "scheduler="))] + ["epochs=1", "model.loss_scale=1e30", "seed=5"]
    tb, _ = _trainer(bad)
    tb.prepare_train()
    before = tb.optimizer.arena.flat.clone()
    tb.train_epoch(0)
    n = len(tb.train_dataloader)
    seen = tb.model._grad_scaler.read()
    assert seen["skipped"] == n and seen["at_floor"] == 0 and abs(seen["scale"] / (1e30 * 0.5 ** n) - 1) < 1e-6
    assert torch.equal(tb.optimizer.arena.flat, before)                      # nothing was applied
    assert not tb.optimizer.exp_avg.any() and not tb.optimizer.exp_avg_sq.any()
    assert f"dropped {n} optimiser step" in capsys.readouterr().out
    assert tb.optimizer.state_dict()["state"] == {}                          # Adam's t is still 0


class FaultyLoader:
    """Fault injection for the fp16 guard's tests, wrapped around the Trainer's loader (global step indices, counted over
    epochs): a poisoned batch carries one +inf target pixel -> a non-finite loss gradient -> the step must be dropped on
    the device; a skipped batch is generated (the random stream advances) but not served -- the clean run the poisoned
    one must equal.  Lives with the tests: the product's loader knows nothing about it."""

    def __init__(self, inner, poison_steps=(), skip_steps=()):
        self.inner, self.poison, self.skip, self.served = inner, set(poison_steps), set(skip_steps), 0

    def __len__(self):
        return len(self.inner)

    def set_epoch(self, epoch):
        self.inner.set_epoch(epoch)

    def __iter__(self):
        for sample in self.inner:
            step, self.served = self.served, self.served + 1
            if step in self.skip:
                continue
            if step in self.poison:
                name = next(k for k in sample if k.endswith("_heatmap"))
                sample[name][0, 0, 0] = float("inf")
            yield sample


def test_fp16_run_with_one_poisoned_batch_equals_the_clean_run_minus_that_step(gpu, run_home, capsys):
    """One batch of a fp16 training run carries an inf target pixel (FaultyLoader above): its backward is
    non-finite, the device-side guard drops exactly that optimiser step and halves the loss scale, and the run ends
    with finite parameters equal to those of the run that never saw the batch -- same weights, same Adam moments, same
    step count.  No host read happens inside the epoch.  The epoch's mean loss leaves exactly that batch out and says so."""
    base = [o for o in OVERRIDES if not o.startswith(("model.dtype=", "epochs=", "gradient_clip=", "scheduler=",
                                                      "eval_epochs=", "save_epochs="))] + \
           ["epochs=1", "scheduler=none", "eval_epochs=null", "save_epochs=null", "train_dataset.length=96"]
    tp, _ = _trainer(base)
    tp.train_dataloader = FaultyLoader(tp.train_dataloader, poison_steps=[4])
    tp.prepare_train()
    tp.train_epoch(0)
    os.chdir(str(run_home))
    tc, _ = _trainer(base)
    tc.train_dataloader = FaultyLoader(tc.train_dataloader, skip_steps=[4])
    tc.prepare_train()
    tc.train_epoch(0)
    sp, sc = tp.model._grad_scaler.read(), tc.model._grad_scaler.read()
    assert sp["skipped"] == 1 and sp["nonfinite"] > 0 and sc["skipped"] == 0
    assert sp["scale"] == sc["scale"] / 2
    printed = capsys.readouterr().out
    assert "dropped 1 optimiser step" in printed and "(1 non-finite losses excluded)" in printed
    assert tp.last_epoch_stats["excluded_losses"] == 1 and tp.last_epoch_stats["dropped_steps"] == 1
    assert np.isfinite(tp.last_epoch_stats["mean_loss"]) and tc.last_epoch_stats["excluded_losses"] == 0
    a, b = tp.optimizer.arena.flat, tc.optimizer.arena.flat
    assert torch.isfinite(a).all() and torch.isfinite(tp.optimizer.exp_avg).all() and torch.isfinite(tp.optimizer.exp_avg_sq).all()
    # the seven later steps ran at half the loss scale: a power of two, so only binary16's subnormal tail can differ
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    assert (tp.optimizer.exp_avg - tc.optimizer.exp_avg).abs().max().item() <= 1e-3 * tc.optimizer.exp_avg.abs().max().item()
    n = len(tp.train_dataloader)
    assert tp.optimizer.step_count == n and tc.optimizer.step_count == n - 1
    assert tp.optimizer.state_dict()["state"][0]["step"].item() == n - 1      # a dropped step does not advance Adam's t
    assert tp.optimizer.state_dict()["bifold_grad_scaler"]["scale"] == sp["scale"]


def test_a_non_finite_loss_without_a_dropped_step_shows_in_the_epoch_mean(gpu, run_home, capsys):
    """bf16 has no scaler and no guarded step: a batch with a non-finite loss is APPLIED, so the epoch's mean must show
    it (the reference's loss.item() mean would), not a finite number over the remaining batches (advisor, round 5)."""
    base = [o for o in OVERRIDES if not o.startswith(("model.dtype=", "epochs=", "gradient_clip=", "scheduler=",
                                                      "eval_epochs=", "save_epochs="))] + \
           ["model.dtype=bf16", "epochs=1", "scheduler=none", "eval_epochs=null", "save_epochs=null",
            "train_dataset.length=48"]
    t, _ = _trainer(base)
    t.train_dataloader = FaultyLoader(t.train_dataloader, poison_steps=[1])
    t.prepare_train()
    t.train_epoch(0)
    assert not np.isfinite(t.last_epoch_stats["mean_loss"]) and t.last_epoch_stats["excluded_losses"] == 0
    assert "non-finite losses excluded" not in capsys.readouterr().out


def test_a_reference_shaped_checkpoint_resumes(gpu, run_home):
    """A `last.pth` written by the reference's Trainer.save_model (torch.optim.Adam state, `epoch` = next epoch,
    `random_states` 4-tuple) loads: start epoch, RNG streams, weights, optimiser moments."""
    import random
    ov = [o for o in OVERRIDES if not o.startswith(("scheduler=", "gradient_clip="))] + ["scheduler=none"]
    t, _ = _trainer(ov)
    t.prepare_train()
    params = [p for p in t.model.parameters() if p.requires_grad]
    ref_opt = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in params], lr=1e-4)
    for p in ref_opt.param_groups[0]["params"]:
        p.grad = torch.full_like(p, 1e-3)
    ref_opt.step()
    random.seed(5); np.random.seed(6); torch.manual_seed(7); torch.cuda.manual_seed(8)
    want_py, want_np, want_t = random.getstate(), np.random.get_state(), torch.get_rng_state()
    want_cuda = torch.cuda.get_rng_state()
    os.makedirs("checkpoints", exist_ok=True)
    torch.save({"epoch": 2, "random_states": (want_py, want_np, want_t, want_cuda),
                "model": t.model.state_dict(), "optimizer": ref_opt.state_dict(), "best_eval": 12.5},
               "checkpoints/last.pth")
    random.seed(0); np.random.seed(0); torch.manual_seed(0); torch.cuda.manual_seed(0)
    t2, _ = _trainer(ov)
    t2.prepare_train()
    assert t2.start_epoch == 2 and t2.metrics.best_eval == 12.5 and t2.optimizer.step_count == 1
    assert random.getstate() == want_py and torch.equal(torch.get_rng_state(), want_t)
    assert torch.equal(torch.cuda.get_rng_state().cpu(), want_cuda.cpu())
    assert np.array_equal(np.random.get_state()[1], want_np[1])
    assert float(t2.optimizer.exp_avg.abs().max()) > 0


def test_torch_clip_grad_norm_on_arena_gradients_equals_torch(gpu, golden_dir):
    """`torch.nn.utils.clip_grad_norm_(model.parameters(), c)` -- the reference's line, unchanged -- on a model whose
    gradients live in the arena: same total norm and same clipped gradients as torch's own implementation; tensors
    that are not arena-backed still go to torch."""
    import bifold.optim as BO
    from test_model_gpu import build_tiny, load_fixture
    from bifold.losses import Losses
    fx = load_fixture(golden_dir, "tiny_unimanual.npz")
    model = build_tiny(False, "fp32", gpu)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fx["state"].items()}, strict=True)
    model = model.to(gpu).train()
    sample = {k: torch.from_numpy(v).to(gpu) for k, v in fx["sample"].items()}
    loss, _ = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))(model(sample), sample)
    loss.backward()
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    copies = [p.detach().clone().requires_grad_(True) for _, p in named]
    for c, (_, p) in zip(copies, named):
        c.grad = p.grad.detach().clone()
    for max_norm in (1e-3, 1e9):                      # clipping active / inactive
        want = BO._torch_clip_grad_norm_(copies, max_norm)
        got = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        assert got.shape == want.shape and abs(got.item() - want.item()) <= 1e-5 * want.item()
        for c, (n, p) in zip(copies, named):
            assert torch.allclose(p.grad, c.grad, rtol=1e-5, atol=1e-12), n
    loose = [torch.nn.Parameter(torch.randn(7, device=gpu)) for _ in range(2)]
    for p in loose:
        p.grad = torch.ones_like(p)
    n = torch.nn.utils.clip_grad_norm_(loose, 1.0)
    assert abs(n.item() - (14 ** 0.5)) < 1e-5 and abs(loose[0].grad[0].item() - 14 ** -0.5) < 1e-5


def test_second_backward_without_step_raises(gpu, golden_dir):
    from test_model_gpu import build_tiny, load_fixture
    from bifold.losses import Losses
    from bifold.optim import Optimizers
    fx = load_fixture(golden_dir, "tiny_unimanual.npz")
    model = build_tiny(False, "fp32", gpu).to(gpu).train()
    sample = {k: torch.from_numpy(v).to(gpu) for k, v in fx["sample"].items()}
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
    opt = Optimizers.get_by_name(dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                                 [p for p in model.parameters() if p.requires_grad])
    loss_fn(model(sample), sample)[0].backward()
    with pytest.raises(RuntimeError, match="gradient accumulation"):
        loss_fn(model(sample), sample)[0].backward()
    opt.zero_grad()
    loss_fn(model(sample), sample)[0].backward()     # consumed -> fine again
    opt.step()
    loss_fn(model(sample), sample)[0].backward()


DDP_SCRIPT = r"""
import os, sys, warnings
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from bifold import dist as bdist
from bifold.losses import Losses
from bifold.models import Models
from bifold.models.clip_encoder import CLIP_CONFIGS
rank, local, world = bdist.init_process_group_from_env()
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", 0)
CLIP_CONFIGS["t"] = dict(embed_dim=32, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
                         context_length=77, vocab_size=512, transformer_width=64, transformer_heads=1, transformer_layers=2)
cfg = dict(name="rgb_clip", image_size=64, is_bimanual=False, requires_graph=False, text_encoder="t", patch_size=16,
           depth=3, heads=2, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
           pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="fp32")
torch.manual_seed(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = Models.get_by_name(cfg, device=dev).to(dev).train()
g = torch.Generator().manual_seed(1)
sample = {"rgb": torch.randn(4, 3, 64, 64, generator=g).to(dev), "instruction": torch.randint(1, 500, (4, 77), generator=g).int().to(dev)}
for k in ("pick", "place"):
    sample[k + "_heatmap"] = torch.rand(4, 64, 64, generator=g).to(dev)
loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
loss_fn(model(sample), sample)[0].backward()
torch.cuda.synchronize()
plain = model.trainable_arena().grad.clone()
model.trainable_arena().grads_pending = False
sync = bdist.gradient_sync_for(model)                # registers on_ready through register_grad_ready_hook
launched = []
orig = sync._launch
sync._launch = lambda spans: (launched.append(list(spans)), orig(spans))[1]
loss_fn(model(sample), sample)[0].backward()         # the real fused backward drives the hooks
sync.finish()
torch.cuda.synchronize()
flat = sorted(sum(launched, []))
assert flat[0][0] == 0 and flat[-1][1] == model.trainable_arena().count
assert all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1)), flat
assert len(launched) >= 2                             # at least one bucket left before the end of backward
got = model.trainable_arena().grad
assert float((got - plain).abs().max()) <= 1e-6 * float(plain.abs().max()), float((got - plain).abs().max())
# fp16 mode over RCCL: the overflow flag of the guarded step rides (MAX, int32) behind the last bucket; a clean
# backward leaves it 0 and the step is applied, a poisoned one (inf target) sets it and the step changes nothing
from bifold.optim import Optimizers
cfg16 = dict(cfg, name="rgb_clip", dtype="fp16")     # (the factory deleted cfg["name"], like the reference's)
torch.manual_seed(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m16 = Models.get_by_name(cfg16, device=dev).to(dev).train()
opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-3, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                             params=[p for p in m16.parameters() if p.requires_grad])
sync16 = bdist.gradient_sync_for(m16)
assert sync16.flag_getter is not None
flags = []
orig_flag = sync16._launch_flag
sync16._launch_flag = lambda: (flags.append(1), orig_flag())[1]
def one_step(s):
    loss_fn(m16(s), s)[0].backward(); sync16.finish(); opt.step(); opt.zero_grad()
before = opt.arena.flat.clone()
one_step(sample)
torch.cuda.synchronize()
assert len(flags) == 1 and m16._grad_scaler.read()["skipped"] == 0 and not torch.equal(opt.arena.flat, before)
bad = dict(sample); bad["pick_heatmap"] = sample["pick_heatmap"].clone(); bad["pick_heatmap"][0, 0, 0] = float("inf")
kept = opt.arena.flat.clone()
one_step(bad)
torch.cuda.synchronize()
seen = m16._grad_scaler.read()
assert len(flags) == 2 and seen["skipped"] == 1 and torch.equal(opt.arena.flat, kept) and torch.isfinite(opt.arena.flat).all()
torch.distributed.destroy_process_group()
print("DDP-HOOKS-OK", len(launched))
"""


def test_backward_hooks_drive_rccl_allreduce_world1(gpu, tmp_path):
    """World-size-1 RCCL group (BIFOLD_FORCE_PG=1) in a child process: the REAL fused backward reports layers through
    register_grad_ready_hook, GradSync all-reduces every arena slice exactly once on its side stream while the
    backward is still running, and the averaged gradients equal the un-synchronised ones.  Then the fp16 mode over the
    same group: the guarded step's overflow flag travels (int32 MAX over RCCL) behind the last bucket, a clean step is
    applied, a poisoned one is dropped with the parameters bit for bit unchanged."""
    script = tmp_path / "ddp_world1.py"
    script.write_text(DDP_SCRIPT)
    env = dict(os.environ, BIFOLD_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", BIFOLD_DDP_COALESCE="1")
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DDP-HOOKS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


DDP2_SCRIPT = """
import os, sys, warnings
import torch
sys.path.insert(0, sys.argv[1])
from bifold import dist as bdist
from bifold.losses import Losses
from bifold.models import Models
from bifold.models.clip_encoder import CLIP_CONFIGS
rank, local_rank, world = bdist.init_process_group_from_env(os.environ["TEST_BACKEND"])
assert world == 2
dev = torch.device("cuda", 0)                       # BOTH ranks on the one GPU of the box
CLIP_CONFIGS["t"] = dict(embed_dim=32, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
                         context_length=77, vocab_size=512, transformer_width=64, transformer_heads=1, transformer_layers=2)
for dtype, tol in (("fp32", 2e-5), ("fp16", 5e-3)):
    cfg = dict(name="rgb_clip", image_size=64, is_bimanual=False, requires_graph=False, text_encoder="t", patch_size=16,
               depth=3, heads=2, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype=dtype)
    torch.manual_seed(0)                            # identical replicas on both ranks
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = Models.get_by_name(cfg, device=dev).to(dev).train()
    g = torch.Generator().manual_seed(1)
    full = {"rgb": torch.randn(8, 3, 64, 64, generator=g).to(dev),
            "instruction": torch.randint(1, 500, (8, 77), generator=g).int().to(dev)}
    for k in ("pick", "place"):
        full[k + "_heatmap"] = torch.rand(8, 64, 64, generator=g).to(dev)
    mine = {k: v[4 * rank:4 * rank + 4].contiguous() for k, v in full.items()}
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
    # the single-process gradient of the whole batch (what the average of the two ranks' gradients must equal)
    loss_fn(model(full), full)[0].backward()
    torch.cuda.synchronize()
    want = model.trainable_arena().grad.clone()
    model.trainable_arena().grads_pending = False
    sync = bdist.gradient_sync_for(model)
    loss_fn(model(mine), mine)[0].backward()        # the real fused backward drives the hooks; buckets leave as layers finish
    sync.finish()
    torch.cuda.synchronize()
    got = model.trainable_arena().grad
    err = float((got - want).abs().max()) / float(want.abs().max())
    assert err < tol, (dtype, err)
    model._grad_hooks.clear()
    print("rank", rank, dtype, "rel err", err, flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("DDP2-OK", rank)
"""


def test_two_ranks_average_gradients_of_the_real_backward(gpu, tmp_path):
    """N > 1 on the data path: TWO ranks (both on this box's one GPU) run the real fused backward on their halves of a
    batch, the hooks send the arena slices through the process group, and every rank ends with the single-process
    gradient of the whole batch -- in the fp32 mode and in the fp16 mode (slices are un-scaled before they leave).
    RCCL is tried first (it may refuse two ranks on one device); gloo on device tensors is the fallback: the
    collective's transport differs, the bucket plan, hooks, stream ordering and loss-scale handling are the product's."""
    import socket
    script = tmp_path / "ddp_world2.py"
    script.write_text(DDP2_SCRIPT)
    tried = {}
    for backend in ("nccl", "gloo"):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        procs = []
        for rank in range(2):
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", RANK=str(rank),
                       LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", TEST_BACKEND=backend, BIFOLD_DDP_COALESCE="1")
            procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                o, e = p.communicate()
            outs.append((p.returncode, o, e))
        tried[backend] = outs
        if all(rc == 0 and "DDP2-OK" in o for rc, o, _ in outs):
            print("two ranks on one GPU over", backend)
            return
    raise AssertionError({b: [(rc, o[-500:], e[-1500:]) for rc, o, e in v] for b, v in tried.items()})
