"""NOT part of the product (SURVEY.md §2 marks metrics OUT OF SCOPE; moved out of the `bifold` package in round 4).
A host-side stand-in for the reference's `bifold.metrics` package, kept as a tool for runs without a reference
checkout: `python -m bifold` itself only tracks the mean annotated-pixel distance for `best.pth` (bifold/__main__.py,
`_TrackedPixelError`) unless `bifold.metrics` is importable.

Only the two families that act on the [B, 2] pixel actions are restated -- `kp_mse` and `ap_<k>`
(bifold/metrics/__init__.py:103-126, 176-213), including their normalisation quirks, because they are
what `tracked_metric: kp_mse` (conf/config.yaml) and checkpoint selection (`best.pth`) depend on.
`iou` (mask head, torchmetrics) and `quantile_prob` (scipy ecdf over whole heatmaps on the host) are
skipped with a notice; plug the reference's package in to get them.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def _valid_and_distance(target: np.ndarray, pred: np.ndarray):
    """target [B, 2] or [B, K, 2] annotated pixels (non-positive = absent), pred [B, 2]."""
    if target.ndim == 3:
        valid = np.min(target, axis=(1, 2)) > 0
        dist = np.linalg.norm(target[valid].round() - pred[valid, None, :], axis=-1).min(axis=1)
    else:
        valid = np.min(target, axis=1) > 0
        dist = np.linalg.norm(target[valid].round() - pred[valid], axis=-1)
    return valid, dist


class _Metric:
    lower_is_better = True

    def __init__(self):
        self.values = []

    def reset(self):
        self.values = []

    def summary(self) -> float:
        return float(np.array(self.values).mean())

    def is_better(self, old_value, new_value) -> bool:
        if old_value is None:
            return True
        return new_value < old_value if self.lower_is_better else new_value > old_value


class KeypointMSE(_Metric):
    def __call__(self, action, sample, **kwargs):
        total, n = 0.0, 0
        for k, pred in action.__dict__.items():
            valid, dist = _valid_and_distance(sample[k].cpu().numpy(), pred)
            total += dist.mean()
            n += valid.sum()
        self.values.append(total / n if n != 0 else 0)          # mean of means over the valid count, as written


class AveragePrecision(_Metric):
    lower_is_better = False

    def __init__(self, threshold: int):
        super().__init__()
        self.threshold = threshold

    def __call__(self, action, sample, **kwargs):
        hit, n = 0, 0
        for k, pred in action.__dict__.items():
            valid, dist = _valid_and_distance(sample[k].cpu().numpy(), pred)
            # correct = within the threshold, or absent and predicted absent (the dummy pick is negative)
            hit += (dist < self.threshold).sum() + (pred[~valid].min(axis=1) < 0).sum()
            n += len(pred)
        self.values.append(hit / n * 100)


class PixelMetrics:
    """Same call surface as bifold.metrics.Metrics: `(action=, sample=, raw_output=)`, `reset()`,
    `summary() -> (has_improved, dict)`, `best_eval`."""

    def __init__(self, cfg):
        self.best_eval: Optional[float] = None
        self.tracked_metric = cfg["tracked_metric"] if isinstance(cfg, dict) else cfg.tracked_metric
        names = cfg["computed_metrics"] if isinstance(cfg, dict) else cfg.computed_metrics
        self.metrics: Dict[str, _Metric] = {}
        self.skipped = []
        for name in names:
            if name == "kp_mse":
                self.metrics[name] = KeypointMSE()
            elif name.startswith("ap_"):
                self.metrics[name] = AveragePrecision(int(name.split("ap_")[-1]))
            elif name in ("iou", "quantile_prob"):
                self.skipped.append(name)
            else:
                raise ValueError(f"Metric {name} not recognized")
        if self.skipped:
            print(f"metrics {self.skipped} need the reference's bifold.metrics package (host-side torchmetrics / "
                  f"scipy); computing {sorted(self.metrics)} only")

    def reset(self):
        for m in self.metrics.values():
            m.reset()

    def __call__(self, *args, **kwargs):
        for m in self.metrics.values():
            m(*args, **kwargs)

    def summary(self):
        improved, out = False, {}
        for name, m in self.metrics.items():
            value = m.summary()
            out[name] = value
            if name == self.tracked_metric and m.is_better(self.best_eval, value):
                self.best_eval, improved = value, True
        return improved, out
