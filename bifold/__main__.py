"""`python -m bifold [group=option ...] [key=value ...]` -- the BiFold trainer entrypoint on MI355X.

Method for method the reference's `Trainer` (bifold/__main__.py:24-325): the same config tree and run
directory (`hydra.run.dir`, keyed by the overrides, chdir into it), the same five factories, the same
step (forward, loss, zero_grad, backward, debug gradient check, `torch.nn.utils.clip_grad_norm_`,
optimizer.step, scheduler.step), the same epoch bookkeeping (periodic pixel evaluation ->
`checkpoints/best.pth`, periodic and final `checkpoints/last.pth`), and the same checkpoint schema
(`epoch` stored as epoch + 1, `random_states` = (python, numpy, torch, cuda), `model`, `optimizer`,
`scheduler`, `best_eval`), so a run started by the reference resumes here and vice versa.

New: under torchrun every rank trains on its own minibatches and gradients are averaged over RCCL,
overlapped with backward (bifold/dist.py); rank 0 alone logs, evaluates and writes checkpoints.
`dataset@train_dataset=synthetic` generates batches in HBM; any other dataset name is served by the
reference's CPU pipeline (`bifold.data`, a namespace portion of this same package when the reference
checkout follows this build on PYTHONPATH -- INTEGRATION.md) and sharded with a DistributedSampler.
Simulator evaluation (`simulator=softgym`) is the reference's `bifold.env` and stays outside this build.
"""
from __future__ import annotations

import os
import random
import sys
import time
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import yaml

from . import dist as bdist
from .config import ConfigNode, compose
from .losses import Losses
from .models import Models
from .optim import Optimizers, Schedulers
from .targets import heatmap_targets

CONF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")


def _head_names(is_bimanual: bool):
    return ([f"{a}_{k}" for a in ("left", "right") for k in ("pick", "place")] if is_bimanual else ["pick", "place"])


class SyntheticLoader:
    """Stands in for the DataLoader: `length // batch_size` batches per epoch of (rgb ~ N(0,1), token
    ids, one annotated pixel per head turned into its target map by bf_gaussian_targets), generated on
    the device with a per-rank, per-epoch seed.  Samples carry the annotated pixels under the head names
    (`pick`, `place`, ...) like the reference's sample dict, which is what the pixel metrics read."""

    def __init__(self, cfg, dataset_cfg, batch_size: int, device, rank: int, world: int, model_name: str,
                 model_cfg, seed_offset: int = 0):
        self.batch = int(batch_size)
        self.size = int(dataset_cfg.image_size)
        self.is_bimanual = bool(dataset_cfg.is_bimanual)
        self.steps = max(1, int(dataset_cfg.length) // (self.batch * world))
        self.device, self.rank, self.seed, self.epoch = device, rank, int(cfg.seed) + seed_offset, 0
        self.sigma = float(cfg.processor.sigma)
        self.strategy = str(cfg.processor.get("strategy", "gmm"))
        self.siglip = model_name.startswith("siglip")
        self.context = int(model_cfg.get("context_length", 0) or 0) if model_name == "siglip_sequential" else 0
        # fault injection for the fp16 guard's tests (global step indices, counted over epochs): a poisoned batch carries
        # one +inf target pixel -> a non-finite loss gradient -> the step must be dropped on the device; a skipped batch
        # is generated (the random stream advances) but not served -- the clean run the poisoned one must equal.
        self.poison_steps = {int(x) for x in (dataset_cfg.get("poison_steps", None) or [])}
        self.skip_steps = {int(x) for x in (dataset_cfg.get("skip_steps", None) or [])}
        self.served = 0

    def __len__(self):
        return self.steps

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed + 1000 * self.epoch + self.rank)
        self.epoch += 1
        s, b, dev = self.size, self.batch, self.device
        names = _head_names(self.is_bimanual)
        for _ in range(self.steps):
            if self.siglip:        # 32 sentencepiece ids + 32 pad (id 1), int64 like the HF processor
                instr = torch.ones(b, 64, dtype=torch.int64, device=dev)
                instr[:, :32] = torch.randint(0, 32000, (b, 32), generator=g, device=dev)
            else:                  # CLIP BPE ids: <start> 30 tokens <end>, zero padded to 77
                instr = torch.zeros(b, 77, dtype=torch.int32, device=dev)
                instr[:, 0] = 49406
                instr[:, 1:31] = torch.randint(1, 49406, (b, 30), generator=g, device=dev).int()
                instr[:, 31] = 49407
            sample = {"rgb": torch.randn(b, 3, s, s, generator=g, device=dev), "instruction": instr,
                      "mask": torch.ones(b, 1, s, s, device=dev)}
            if self.context:
                sample["rgb_context"] = torch.randn(b, self.context, 3, s, s, generator=g, device=dev)
                k = torch.randint(0, self.context + 1, (b, 1), generator=g, device=dev)
                sample["context_attention_mask"] = (torch.arange(self.context, device=dev)[None, :] < k).long()
            pixels = {n: torch.randint(1, s, (b, 2), generator=g, device=dev).float() for n in names}
            sample.update(pixels)
            sample.update(heatmap_targets(pixels, s, self.sigma, self.strategy))
            step, self.served = self.served, self.served + 1
            if step in self.skip_steps:
                continue
            if step in self.poison_steps:
                sample[f"{names[0]}_heatmap"][0, 0, 0] = float("inf")
            yield sample


def shard_dataloader(dl, rank: int, world: int, seed: int):
    """The reference builds one shuffled DataLoader (data/__init__.py:67-72); under data parallelism every
    rank would draw the SAME batches from it.  Rebuild it over the same dataset with a DistributedSampler
    (disjoint shards, reshuffled per epoch through `sampler.set_epoch`)."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    sampler = DistributedSampler(dl.dataset, num_replicas=world, rank=rank, shuffle=True, seed=int(seed))
    return DataLoader(dl.dataset, batch_size=dl.batch_size, sampler=sampler, num_workers=dl.num_workers,
                      collate_fn=dl.collate_fn, drop_last=dl.drop_last, pin_memory=True)


class Trainer:
    def __init__(self, cfg: ConfigNode, overrides=()):
        self.cfg = cfg
        self.rank, self.local_rank, self.world = (bdist.init_process_group_from_env() if cfg.get("ddp", True)
                                                  else (0, 0, 1))
        if not torch.cuda.is_available() or cfg.use_cpu:
            raise RuntimeError("this build of bifold trains on the MI355X only; the CPU restatement of the path "
                               "lives in oracle/ and is test infrastructure, not a fallback")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.writer = self._make_writer(overrides)
        self.seed_randomness()
        self.model_name = str(cfg.model.name)          # the factory deletes cfg.model.name, like the reference
        model_cfg = ConfigNode(cfg.model)
        self.model = Models.get_by_name(cfg.model, device=self.device).to(self.device)
        self.train_dataloader, self.test_dataloader, self.input_processor = self._make_loaders(model_cfg)
        self.metrics = self._make_metrics()
        self.sync = None
        self.start_epoch = 0

    # ------------------------------------------------------------------ construction helpers
    def _make_writer(self, overrides):
        cfg = self.cfg
        if not (cfg.use_wandb and not cfg.eval_only and not cfg.debug) or self.rank != 0:
            return None
        try:
            import wandb
        except ImportError:
            print("use_wandb is set but wandb is not installed: logging to stdout only")
            return None
        return wandb.init(project="bifold", group=cfg.train_dataset.name, name="+".join(overrides), resume="allow")

    def _make_loaders(self, model_cfg):
        cfg = self.cfg
        name = cfg.train_dataset.name
        if name == "synthetic":
            if cfg.test_dataset.name is None:
                cfg.test_dataset = cfg.train_dataset       # data/__init__.py:74-75
            train = None if cfg.eval_only else SyntheticLoader(cfg, cfg.train_dataset, cfg.batch_size, self.device,
                                                               self.rank, self.world, self.model_name, model_cfg)
            test = SyntheticLoader(cfg, cfg.test_dataset, cfg.test_batch_size, self.device, 0, 1, self.model_name,
                                   model_cfg, seed_offset=500_000)
            test.steps = max(1, min(test.steps, int(cfg.test_dataset.get("test_length", 64)) // test.batch))
            return train, test, None
        try:
            from bifold.data import Datasets            # the reference's CPU pipeline, if on the path
        except ImportError as e:
            raise NotImplementedError(
                f"dataset {name!r} needs the reference's bifold.data package on PYTHONPATH behind this build "
                f"(INTEGRATION.md); only `synthetic` is built in") from e
        train_dl, test_dl, processor = Datasets.get_dataloaders(cfg)
        if self.world > 1 and train_dl is not None:
            train_dl = shard_dataloader(train_dl, self.rank, self.world, cfg.seed)
        return train_dl, test_dl, processor

    def _make_metrics(self):
        try:
            from bifold.metrics import Metrics          # the reference's, if on the path
            return Metrics(self.cfg.metrics)
        except ImportError:
            from .pixel_metrics import PixelMetrics
            return PixelMetrics(self.cfg.metrics)

    # ------------------------------------------------------------------ reference Trainer surface
    def train(self):
        if self.start_epoch < self.cfg.epochs:
            for epoch in range(self.start_epoch, self.cfg.epochs):
                self.train_epoch(epoch)
                if self.cfg.eval_epochs and (epoch + 1) % self.cfg.eval_epochs == 0:
                    has_improved, _ = self.eval_epoch(epoch)
                    if has_improved:
                        self.save_model(epoch, is_best=True)
                if self.cfg.save_epochs and (epoch + 1) % self.cfg.save_epochs == 0:
                    self.save_model(epoch)
            self.save_model(self.cfg.epochs - 1)
            if self.world > 1:
                torch.distributed.barrier()       # rank 0's last.pth is complete before anyone re-reads it
        elif self.rank == 0:
            print(f"checkpoint is already at epoch {self.start_epoch} >= epochs={self.cfg.epochs}: nothing to train")

    def eval(self):
        self.load_model(load_best=self.cfg.load_best)
        eval_file = f"eval_{self.cfg.test_dataset.name}.yaml"
        _, metric_dict = self.eval_epoch()
        if self.rank != 0:
            return metric_dict
        for k, val in metric_dict.items():
            if isinstance(val, dict):
                for sub_k, sub_val in val.items():
                    print(f"{k} {sub_k}:\t{sub_val:.2f}")
            else:
                print(f"{k}:\t{val:.2f}")
        if os.path.isfile(eval_file):
            with open(eval_file) as f:
                old = yaml.load(f, Loader=yaml.Loader) or {}
            for k, val in old.items():
                if k not in metric_dict:
                    metric_dict[k] = val
                elif val != metric_dict[k]:
                    print(f"Old value for {k} = {val} ; New value {metric_dict[k]}")
        with open(eval_file, "w") as f:
            yaml.dump(metric_dict, f)
        return metric_dict

    def seed_randomness(self):
        random.seed(self.cfg.seed)
        np.random.seed(self.cfg.seed)
        torch.manual_seed(self.cfg.seed)
        torch.cuda.manual_seed_all(self.cfg.seed)

    def prepare_train(self):
        params_non_frozen = filter(lambda p: p.requires_grad, self.model.parameters())
        self.loss_fn = Losses.get_by_name(cfg=self.cfg.loss)
        self.optimizer = Optimizers.get_by_name(cfg=self.cfg.optim, params=params_non_frozen)
        assert self.train_dataloader is not None
        self.scheduler = Schedulers.get_by_name(cfg=self.cfg.scheduler, optimizer=self.optimizer,
                                                max_iters=len(self.train_dataloader) * self.cfg.epochs)
        if self.world > 1 or os.environ.get("BIFOLD_FORCE_PG") == "1":
            self.sync = bdist.gradient_sync_for(self.model)
        self.load_model()

    def train_epoch(self, epoch=None):
        self.model.train()
        dl = self.train_dataloader
        if epoch is not None:                     # per-epoch reshuffle / reseed, also after a resume
            if hasattr(dl, "set_epoch"):
                dl.set_epoch(epoch)
            elif hasattr(getattr(dl, "sampler", None), "set_epoch"):
                dl.sampler.set_epoch(epoch)
        total, t0 = None, time.time()
        for i, sample in enumerate(dl):
            sample = self.move_sample_to_device(sample)
            output = self.model(sample)
            loss, intermediate_losses = self.loss_fn(output, sample)
            self.optimizer.zero_grad()
            assert loss is not None
            loss.backward()
            if self.cfg.debug:
                for name, param in self.model.named_parameters():
                    if param.requires_grad and param.grad is None:
                        raise ValueError(f"Parameter {name} might not have gradient attached!")
            if self.sync is not None:
                self.sync.finish()                # averaged gradients before they are clipped / applied
            if self.cfg.gradient_clip is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.cfg.gradient_clip)
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
            if self.writer:
                self.writer.log({"loss": loss, "epoch": epoch})
                for k, val in intermediate_losses.items():
                    self.writer.log({k: val})
                for j, param_group in enumerate(self.optimizer.param_groups):
                    self.writer.log({f"lr_{j}": param_group["lr"]})
            # the reference calls loss.item() every step (a device sync); keep a device-side running sum
            # instead and read it once per epoch
            total = loss.detach() if total is None else total + loss.detach()
        n = len(dl)
        if self.rank == 0:
            mean = float(total.item()) / n
            dt = time.time() - t0
            print(f"epoch {epoch}: mean loss {mean:.6f}  {n * self.cfg.batch_size * self.world / dt:.1f} samples/s "
                  f"lr {self.optimizer.param_groups[0]['lr']:.3e}")
        tower = getattr(self.model, "clip_encoder", None) or getattr(self.model, "siglip_model", None)
        if tower is not None:
            tower.check_token_ids()
        scaler = getattr(self.model, "_grad_scaler", None)        # fp16 mode: guarded step (models/gradscale.py)
        if scaler is not None:
            # one device read per epoch, like the loss above.  A backward that overflowed binary16 was DROPPED on the
            # device (parameters and moments untouched, loss scale halved) -- nothing to repair here; the counters are
            # identical on every rank (the flag is MAX-reduced), so all ranks report / raise together.
            seen = scaler.read()
            dropped = scaler.overflowed()
            if dropped and self.rank == 0:
                print(f"epoch {epoch}: model.dtype=fp16 dropped {dropped} optimiser step(s) with non-finite gradients; "
                      f"loss scale now {seen['scale']:g}")
            if seen["at_floor"]:
                raise FloatingPointError(f"model.dtype=fp16: gradients overflow even at loss scale {seen['scale']:g} "
                                         f"({seen['at_floor']} step(s) in or before epoch {epoch}): the run diverged, or "
                                         f"needs model.dtype=bf16")

    def eval_epoch(self, epoch=None) -> Tuple[Optional[bool], Dict]:
        self.model.eval()
        has_improved = None
        if epoch is not None or self.cfg.simulator is None:
            has_improved, metric_dict = self.eval_epoch_pixel(epoch)
        elif self.cfg.simulator == "softgym":
            # SoftGym / PyFlex rollouts are the reference's bifold.env (CUDA-only FleX libraries): outside this
            # build.  Said loudly, and the run still ends cleanly after training (checkpoints are written).
            if self.rank == 0:
                print("simulator=softgym: simulator rollouts need the reference's bifold.env package and PyFlex; "
                      "SKIPPING the final simulator evaluation (pass simulator=null for the pixel metrics)")
            metric_dict = {}
        else:
            raise ValueError(f"Simulator {self.cfg.simulator} not supported")
        if self.writer:
            for k, val in metric_dict.items():
                self.writer.log({k: val})
        if epoch is not None:
            assert has_improved is not None, "has_improved boolean must be specified during training"
        return has_improved, metric_dict

    def eval_epoch_pixel(self, epoch):
        """bifold/__main__.py:189-238: `get_action` on every test batch (arg-max + mask snap on the device,
        [B, 2] pixels to the host) feeding the pixel metrics.  Every rank runs it on the same test set, so
        `has_improved` agrees across ranks without a broadcast."""
        self.metrics.reset()
        for sample in self.test_dataloader:
            sample = self.move_sample_to_device(sample)
            action, raw_output = self.model.get_action(sample, return_raw_output=True)
            if any("pick" in k for k in sample.keys()):
                self.metrics(action=action, sample=sample, raw_output=raw_output)
            if self.cfg.visualize_predictions:
                raise NotImplementedError("visualize_predictions uses the reference's bifold.utils.visualization "
                                          "(matplotlib / cv2, host side)")
        return self.metrics.summary()

    # ------------------------------------------------------------------ checkpoints (reference schema)
    def load_model(self, load_best=False):
        if load_best and os.path.isfile(os.path.join("checkpoints", "best.pth")):
            checkpoint_file = os.path.join("checkpoints", "best.pth")
        elif os.path.isfile(os.path.join("checkpoints", "last.pth")):
            checkpoint_file = os.path.join("checkpoints", "last.pth")
        else:
            if self.cfg.eval_only and not self.cfg.debug:
                raise FileNotFoundError("Cannot evaluate with untrained model")
            self.start_epoch = 0
            return
        checkpoint = torch.load(checkpoint_file, map_location=self.device, weights_only=False)
        self.start_epoch = checkpoint["epoch"]
        states = checkpoint["random_states"]
        random.setstate(states[0])
        np.random.set_state(states[1])
        torch.set_rng_state(states[2].cpu())
        if states[3] is not None:
            torch.cuda.set_rng_state(states[3].cpu())
        self.model.load_state_dict(checkpoint["model"])
        if not self.cfg.eval_only:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
            if self.scheduler is not None and "scheduler" in checkpoint:
                self.scheduler.load_state_dict(checkpoint["scheduler"])
            if "best_eval" in checkpoint:
                self.metrics.best_eval = checkpoint["best_eval"]
        if self.rank == 0:
            print(f"Loaded model from checkpoint {checkpoint_file}")

    def save_model(self, epoch, is_best=False):
        if self.rank != 0:
            return
        os.makedirs("checkpoints", exist_ok=True)
        model_path = os.path.join("checkpoints", "best.pth" if is_best else "last.pth")
        state_dict = {
            "epoch": epoch + 1,
            "random_states": (random.getstate(), np.random.get_state(), torch.get_rng_state(),
                              torch.cuda.get_rng_state() if torch.cuda.is_available() else None),
            "model": self.model.state_dict(),
            "optimizer": self.optimizer.state_dict(),
        }
        if self.scheduler is not None:
            state_dict["scheduler"] = self.scheduler.state_dict()
        if self.metrics.best_eval is not None:
            state_dict["best_eval"] = self.metrics.best_eval
        torch.save(state_dict, model_path)

    def move_sample_to_device(self, sample):
        for k, val in sample.items():
            if isinstance(val, torch.Tensor):
                sample[k] = val.to(self.device, non_blocking=True)
        return sample


def run(cfg: ConfigNode, overrides=()) -> Trainer:
    """Body of the reference's `main(cfg)` (bifold/__main__.py:24-35): dump the composed config into the
    run directory, train unless `eval_only`, then evaluate."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(os.getcwd())
        with open("config.yaml", "w") as f:
            yaml.safe_dump(_plain(cfg), f)
    trainer = Trainer(cfg, overrides)
    if not cfg.eval_only:
        trainer.prepare_train()
        trainer.train()
    trainer.eval()
    return trainer


def _plain(node):
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    return node


def main(argv=None) -> Trainer:
    """`@hydra.main(config_path="conf", config_name="config")` without hydra-core (absent from this image):
    compose conf/ with the overrides, create `hydra.run.dir` and chdir into it (`hydra.job.chdir: true`), so
    `checkpoints/last.pth` is found where the reference's Hydra run would have left it."""
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg, hydra = compose(CONF_DIR, argv, return_hydra=True)
    run_dir = os.path.expanduser(str(hydra.run.dir))
    os.makedirs(run_dir, exist_ok=True)
    if hydra.job.chdir:
        os.chdir(run_dir)
    return run(cfg, argv)


if __name__ == "__main__":
    main()
