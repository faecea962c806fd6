"""Minimal Hydra-style config composition for `python -m bifold`.

The reference's entrypoint is `@hydra.main(config_path="conf", config_name="config")`
(bifold/__main__.py:24).  hydra-core/omegaconf are used when installed; this module is the
fallback for images without them (this one), covering exactly what the conf tree uses:
`defaults:` lists with `group: option` and `group@package: option` entries, `group=option`
and `key.path=value` command-line overrides, `${a.b}` / `${.sibling}` / `${oc.env:VAR}`
interpolation, and the `hydra:` block of conf/config.yaml (run directory keyed by
`hydra.job.override_dirname`, `job.chdir`), so a run lands in the directory the reference's
Hydra would have picked and auto-resume finds the same `checkpoints/last.pth`.  Nodes behave like the OmegaConf nodes the factories expect: attribute and
item access, `del cfg.name`, and `**cfg` splatting.
"""
from __future__ import annotations

import os
import re
from typing import Any, Dict, List, Optional

import yaml


class ConfigNode(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        try:
            del self[key]
        except KeyError:
            raise AttributeError(key) from None


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigNode({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    return obj


def _load_yaml(path: str):
    with open(path) as f:
        # `1e-4` is a string for PyYAML's default resolver; accept it as a float like OmegaConf does
        loader = yaml.SafeLoader
        data = yaml.load(f, Loader=loader)
    return _fix_floats(data if data is not None else {})


_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)$")


def _fix_floats(obj):
    if isinstance(obj, dict):
        return {k: _fix_floats(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_fix_floats(v) for v in obj]
    if isinstance(obj, str) and _FLOAT.match(obj):
        return float(obj)
    return obj


def _set_path(root: Dict, path: str, value) -> None:
    parts = path.split(".")
    node = root
    for p in parts[:-1]:
        node = node.setdefault(p, {})
    node[parts[-1]] = value


def _get_path(root: Dict, path: str):
    node = root
    for p in path.split("."):
        node = node[p]
    return node


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve(root: Dict) -> None:
    def lookup(expr: str, here: str):
        if expr.startswith("oc.env:"):
            name, _, default = expr[7:].partition(",")
            val = os.environ.get(name.strip(), default.strip() if default else None)
            if val is None:
                raise KeyError(f"environment variable {name} is not set")
            return val
        if expr.startswith("."):                       # relative to the containing node
            base = here.rsplit(".", 1)[0] if "." in here else ""
            expr = (base + "." if base else "") + expr.lstrip(".")
        return resolve_value(_get_path(root, expr), expr)

    def resolve_value(val, here: str):
        if isinstance(val, str):
            m = _INTERP.fullmatch(val)
            if m:
                return lookup(m.group(1), here)
            for _ in range(16):
                if not _INTERP.search(val):
                    break
                val = _INTERP.sub(lambda mm: str(lookup(mm.group(1), here)), val)
            return val
        return val

    def walk(node, prefix: str):
        items = node.items() if isinstance(node, dict) else enumerate(node)
        for k, v in list(items):
            here = f"{prefix}.{k}" if prefix else str(k)
            if isinstance(v, (dict, list)):
                walk(v, here)
            else:
                node[k] = resolve_value(v, here)

    walk(root, "")


def override_dirname(overrides: List[str], exclude_keys=(), item_sep: str = ",", kv_sep: str = "=") -> str:
    """Hydra's `hydra.job.override_dirname`: the task overrides whose key (or config group, for
    `group@package=option`) is not excluded, sorted, joined by `item_sep`."""
    lines = []
    for ov in overrides:
        key = ov.partition("=")[0].lstrip("+~")
        group = key.partition("@")[0]
        if key in exclude_keys or group in exclude_keys:
            continue
        lines.append(ov)
    lines.sort()
    return item_sep.join(lines).replace("=", kv_sep)


def compose(config_dir: str, overrides: Optional[List[str]] = None, config_name: str = "config",
            return_hydra: bool = False):
    """The task config (without the `hydra` node, as Hydra hands it to the task function); with
    `return_hydra` also the resolved `hydra` node (`run.dir`, `job.chdir`, `job.override_dirname`,
    `overrides.task`)."""
    overrides = list(overrides or [])
    base = _load_yaml(os.path.join(config_dir, config_name + ".yaml"))
    defaults = base.pop("defaults", [])
    groups = []                                          # (group, package, option)
    for entry in defaults:
        if entry == "_self_":
            continue
        (key, option), = entry.items()
        if key.startswith("override "):
            continue                                     # hydra-internal (logging) overrides
        group, _, package = key.partition("@")
        groups.append([group, package or group, option])
    values = []
    for ov in overrides:
        key, sep, val = ov.partition("=")
        if not sep:
            raise ValueError(f"override {ov!r} is not key=value")
        key = key.lstrip("+")
        hit = [g for g in groups if key in (g[0] + "@" + g[1], g[1]) or (key == g[0] and g[0] == g[1])]
        if hit:
            for g in hit:
                g[2] = val
        else:
            values.append((key, _fix_floats(yaml.safe_load(val))))
    cfg: Dict[str, Any] = dict(base)
    for group, package, option in groups:
        if option in (None, "null"):
            continue
        path = os.path.join(config_dir, group, f"{option}.yaml")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"config group {group!r} has no option {option!r} ({path})")
        cfg[package] = _load_yaml(path)
    for key, val in values:
        _set_path(cfg, key, val)
    hydra = cfg.setdefault("hydra", {}) or {}
    cfg["hydra"] = hydra
    job = hydra.setdefault("job", {})
    od = ((job.get("config") or {}).get("override_dirname") or {})
    job["override_dirname"] = override_dirname(overrides, tuple(od.get("exclude_keys") or ()),
                                               od.get("item_sep", ","), od.get("kv_sep", "="))
    job.setdefault("chdir", False)
    hydra.setdefault("run", {}).setdefault("dir", ".")
    hydra["overrides"] = {"task": list(overrides)}
    _resolve(cfg)
    hydra = _wrap(cfg.pop("hydra"))
    out = _wrap(cfg)
    return (out, hydra) if return_hydra else out
