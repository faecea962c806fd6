"""`python -m bifold_amd [group=option ...] [key=value ...]` -- the BiFold trainer entrypoint
on MI355X.

Same config tree, factories and step semantics as the reference trainer
(bifold/__main__.py:24-163, 272-325): seed -> Models.get_by_name(cfg.model).to(device) ->
Losses / Optimizers / Schedulers -> per batch: forward, loss, zero_grad, backward,
(debug: every trainable parameter must have a gradient), optional grad-norm clip,
optimizer.step, scheduler.step; checkpoints carry the same keys (`epoch`, RNG states,
`model`, `optimizer`, `scheduler`, `best_eval`).  New: under torchrun every rank trains on
its own minibatches and gradients are averaged over RCCL, overlapped with backward
(bifold_amd/dist.py); rank 0 alone logs and writes checkpoints.

The reference's data pipeline, metrics and simulator evaluation are CPU-side and outside
this build's hot path: `dataset@train_dataset=synthetic` generates batches in HBM, and any
other dataset name is delegated to the reference's `bifold.data` when that package is
importable.
"""
from __future__ import annotations

import os
import random
import sys
import time

import numpy as np
import torch

from . import dist as bdist
from .config import ConfigNode, compose
from .losses import Losses
from .models import Models
from .optim import Optimizers, Schedulers
from .targets import heatmap_targets

CONF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")


class SyntheticLoader:
    """Stands in for the DataLoader: `length // batch_size` batches per epoch of
    (rgb ~ N(0,1), CLIP-style token ids, one annotated pixel per head turned into its target
    map by bf_gaussian_targets), generated on the device with a per-rank, per-epoch seed."""

    def __init__(self, cfg, device, rank: int, world: int, model_name: str):
        self.batch = int(cfg.batch_size)
        self.size = int(cfg.train_dataset.image_size)
        self.is_bimanual = bool(cfg.train_dataset.is_bimanual)
        self.steps = max(1, int(cfg.train_dataset.length) // (self.batch * world))
        self.device, self.rank, self.seed, self.epoch = device, rank, int(cfg.seed), 0
        self.sigma = float(cfg.processor.sigma)
        self.strategy = str(cfg.processor.get("strategy", "gmm"))
        self.siglip = model_name.startswith("siglip")
        self.context = int(cfg.model.get("context_length", 0) or 0) if model_name == "siglip_sequential" else 0

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed + 1000 * self.epoch + self.rank)
        self.epoch += 1
        s, b, dev = self.size, self.batch, self.device
        names = ([f"{a}_{k}" for a in ("left", "right") for k in ("pick", "place")]
                 if self.is_bimanual else ["pick", "place"])
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
            pixels = {n: torch.randint(0, s, (b, 2), generator=g, device=dev).float() for n in names}
            sample.update(heatmap_targets(pixels, s, self.sigma, self.strategy))
            yield sample


class Trainer:
    def __init__(self, cfg: ConfigNode):
        self.cfg = cfg
        self.rank, self.local_rank, self.world = bdist.init_process_group_from_env() if cfg.ddp else (0, 0, 1)
        if not torch.cuda.is_available() or cfg.use_cpu:
            raise RuntimeError("bifold_amd trains on the MI355X only; the CPU restatement of the path "
                               "lives in oracle/ and is test infrastructure, not a fallback")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.seed_randomness()
        self.model_name = str(cfg.model.name)          # the factory deletes cfg.model.name, like the reference
        self.model = Models.get_by_name(cfg.model, device=self.device).to(self.device)
        self.train_dataloader = self._make_loader()
        self.sync = None
        self.start_epoch = 0
        self.best_eval = None

    def _make_loader(self):
        name = self.cfg.train_dataset.name
        if name == "synthetic":
            return SyntheticLoader(self.cfg, self.device, self.rank, self.world, self.model_name)
        try:
            from bifold.data import Datasets            # the reference's CPU pipeline, if installed
        except ImportError as e:
            raise NotImplementedError(
                f"dataset {name!r} needs the reference's bifold.data package next to this build "
                f"(see INTEGRATION.md); only `synthetic` is built in") from e
        train_dl, _, _ = Datasets.get_dataloaders(self.cfg)
        return train_dl

    def seed_randomness(self):
        random.seed(self.cfg.seed)
        np.random.seed(self.cfg.seed)
        torch.manual_seed(self.cfg.seed)
        torch.cuda.manual_seed_all(self.cfg.seed)

    def prepare_train(self):
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.loss_fn = Losses.get_by_name(cfg=self.cfg.loss)
        self.optimizer = Optimizers.get_by_name(cfg=self.cfg.optim, params=params)
        self.scheduler = Schedulers.get_by_name(cfg=self.cfg.scheduler, optimizer=self.optimizer,
                                                max_iters=len(self.train_dataloader) * self.cfg.epochs)
        if self.world > 1:
            self.sync = bdist.gradient_sync_for(self.model)
        self.load_model()

    def move_sample_to_device(self, sample):
        return {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v)
                for k, v in sample.items()}

    def train_epoch(self, epoch=None):
        self.model.train()
        total, t0 = None, time.time()
        for i, sample in enumerate(self.train_dataloader):
            sample = self.move_sample_to_device(sample)
            output = self.model(sample)
            loss, intermediate = self.loss_fn(output, sample)
            self.optimizer.zero_grad()
            loss.backward()
            if self.cfg.debug:
                for name, p in self.model.named_parameters():
                    if p.requires_grad and p.grad is None:
                        raise ValueError(f"Parameter {name} might not have gradient attached!")
            if self.sync is not None:
                self.sync.finish()
            if self.cfg.gradient_clip is not None:
                self.optimizer.clip_grad_norm_(self.cfg.gradient_clip)
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
            # the reference calls loss.item() every step (a device sync); keep a device-side
            # running sum instead and read it once per epoch
            total = loss.detach() if total is None else total + loss.detach()
        n = len(self.train_dataloader)
        if self.rank == 0:
            mean = float(total.item()) / n
            dt = time.time() - t0
            print(f"epoch {epoch}: mean loss {mean:.6f}  {n * self.cfg.batch_size * self.world / dt:.1f} samples/s "
                  f"lr {self.optimizer.param_groups[0]['lr']:.3e}")
        tower = getattr(self.model, "clip_encoder", None) or getattr(self.model, "siglip_model", None)
        if tower is not None:
            tower.check_token_ids()

    def train(self):
        for epoch in range(self.start_epoch, self.cfg.epochs):
            self.train_epoch(epoch)
            if self.cfg.save_epochs and (epoch + 1) % self.cfg.save_epochs == 0:
                self.save_model(epoch)
        if self.start_epoch < self.cfg.epochs:
            self.save_model(self.cfg.epochs - 1)

    # -- checkpoints: same layout as bifold/__main__.py:272-325 ---------------------------
    def _ckpt_path(self, best=False):
        d = os.path.join(os.path.expanduser(str(self.cfg.output_dir)), "checkpoints")
        return os.path.join(d, "best.pth" if best else "last.pth")

    def load_model(self, load_best=False):
        path = self._ckpt_path(load_best)
        if not os.path.isfile(path):
            if self.cfg.eval_only and not self.cfg.debug:
                raise FileNotFoundError(f"No checkpoint found at {path}")
            return
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.start_epoch = ck["epoch"] + 1
        random.setstate(ck["python_rng_state"])
        np.random.set_state(ck["numpy_rng_state"])
        torch.set_rng_state(ck["torch_rng_state"].cpu())
        self.model.load_state_dict(ck["model"])
        if hasattr(self, "optimizer"):
            self.optimizer.load_state_dict(ck["optimizer"])
            if self.scheduler is not None and "scheduler" in ck:
                self.scheduler.load_state_dict(ck["scheduler"])
        self.best_eval = ck.get("best_eval")

    def save_model(self, epoch, is_best=False):
        if self.rank != 0:
            return
        path = self._ckpt_path(is_best)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        ck = {"epoch": epoch, "python_rng_state": random.getstate(), "numpy_rng_state": np.random.get_state(),
              "torch_rng_state": torch.get_rng_state(), "cuda_rng_state": torch.cuda.get_rng_state_all(),
              "model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
              "best_eval": self.best_eval}
        if self.scheduler is not None:
            ck["scheduler"] = self.scheduler.state_dict()
        torch.save(ck, path)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = compose(CONF_DIR, argv)
    trainer = Trainer(cfg)
    if not cfg.eval_only:
        trainer.prepare_train()
        trainer.train()
    else:
        raise NotImplementedError("pixel / simulator evaluation lives in the reference's bifold.metrics and "
                                  "bifold.env packages (CPU, outside this build's hot path)")


if __name__ == "__main__":
    main()
