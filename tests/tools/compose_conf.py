"""Compose the reference's Hydra ``conf/`` tree with PyYAML alone (neither hydra nor omegaconf is installed here) and resolve the
``${...}`` interpolations of the three plug-in subtrees -- what ``MetaDetector.__post_init__`` hands to ``instantiate``
(``/root/reference/src/torchbox3d/nn/meta/arch.py:41-46``): ``model._backbone``, ``model._head``, ``model._decoder``.

Only the Hydra semantics this tree uses are implemented:
  * defaults lists: ``- name`` (same group, package of the including file), ``- /group: name`` and ``- override /group: name``
    (absolute group, package = group), ``_self_``; a later entry overrides an earlier one; ``???`` options must be overridden;
  * ``# @package _global_`` headers (first line) -- a file without one that is pulled in by a bare ``- name`` entry inherits the
    including file's package;
  * OmegaConf interpolations: absolute ``${a.b}``, relative ``${.x}`` / ``${..x}`` (relative to the node that HOLDS the key),
    list indices (``${.layers.0}``); custom resolvers (``${oc.env:..}``, ``${hydra:..}``, ``${now:..}``) are left as strings --
    none is reachable from the three subtrees.

    python tests/tools/compose_conf.py            # rewrites tests/golden/conf_kwargs.json (data: resolved kwargs, not YAML text)
"""
from __future__ import annotations

import copy
import json
import math
import os
import re
import sys
from typing import Any, Dict, List, Tuple

import yaml

CONF = "/root/reference/conf"
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(os.path.dirname(HERE), "golden", "conf_kwargs.json")

# the three `_target_`s INTEGRATION.md section 2 tells a maintainer to change (everything else stays as the reference ships it)
TARGET_SWAP = {
    "torchbox3d.nn.backbones.dla.RangeNet": "range_view_3d_detection_amd.nn.backbones.dla.RangeNet",
    "torchbox3d.nn.heads.detection_head.DetectionHead": "range_view_3d_detection_amd.nn.heads.detection_head.DetectionHead",
    "torchbox3d.nn.decoders.range_decoder.RangeDecoder": "range_view_3d_detection_amd.nn.decoders.range_decoder.RangeDecoder",
}


def _load(group: str, name: str) -> Tuple[Dict[str, Any], bool]:
    path = os.path.join(CONF, group, name + ".yaml")
    with open(path) as f:
        text = f.read()
    is_global = text.lstrip().startswith("# @package _global_")
    return yaml.safe_load(text) or {}, is_global


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _place(cfg: Dict[str, Any], package: str) -> Dict[str, Any]:
    out: Dict[str, Any] = cfg
    for part in reversed([p for p in package.split(".") if p]):
        out = {part: out}
    return out


def _collect_overrides(group: str, name: str, overrides: Dict[str, str]) -> None:
    """First pass (Hydra resolves the whole defaults tree before composing): `override /group: option` anywhere below wins."""
    cfg, _ = _load(group, name)
    for entry in cfg.get("defaults", []):
        if isinstance(entry, str):
            if entry != "_self_":
                _collect_overrides(group, entry, overrides)
            continue
        ((key, option),) = entry.items()
        if key.startswith("override "):
            overrides[key[len("override "):].strip().lstrip("/")] = option
        elif option != "???":
            _collect_overrides(key.lstrip("/"), option, overrides)


def _compose(group: str, name: str, package: str, overrides: Dict[str, str]) -> Dict[str, Any]:
    cfg, is_global = _load(group, name)
    if is_global:
        package = ""
    defaults: List[Any] = cfg.pop("defaults", [])
    if "_self_" not in defaults:
        defaults = ["_self_"] + list(defaults)  # (Hydra 1.1+: a primary config without _self_ is composed FIRST)
    out: Dict[str, Any] = {}
    for entry in defaults:
        if entry == "_self_":
            _merge(out, _place(cfg, package))
        elif isinstance(entry, str):
            _merge(out, _compose(group, entry, package, overrides))
        else:
            ((key, option),) = entry.items()
            if key.startswith("override "):
                continue  # (applied where the group is first listed)
            g = key.lstrip("/")
            if g == "hydra/launcher" or g.startswith("hydra"):
                continue
            option = overrides.get(g, option)
            if option == "???":
                raise ValueError(f"config group {g!r} has no option (??? in {group}/{name}.yaml)")
            _merge(out, _compose(g, option, g.replace("/", "."), overrides))
    return out


def compose(experiment: str) -> Dict[str, Any]:
    """The composed, UNRESOLVED global config of `+experiment=<experiment>` (conf/config.yaml:10-13)."""
    overrides: Dict[str, str] = {}
    _collect_overrides("experiment", experiment, overrides)
    return _compose("experiment", experiment, "", overrides)


_INTERP = re.compile(r"\$\{([^${}]+)\}")
_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")


def _lookup(root: Any, path: List[Any]) -> Any:
    node = root
    for p in path:
        if isinstance(node, list):
            node = node[int(p)]
        elif p in node:
            node = node[p]
        elif isinstance(p, str) and p.lstrip("-").isdigit() and int(p) in node:
            node = node[int(p)]
        else:
            raise KeyError(path)
    return node


def _resolve_value(root: Dict[str, Any], holder: List[Any], value: Any, depth: int = 0) -> Any:
    """`holder` = path of the container that holds the key whose value this is."""
    if depth > 32:
        raise RecursionError("interpolation cycle")
    if isinstance(value, dict):
        return value  # (resolved by the walker)
    if not isinstance(value, str):
        return value
    m = _INTERP.fullmatch(value.strip())
    if not m:
        if _FLOAT.match(value) and not value.isdigit():  # PyYAML reads `1e-3` as a string; OmegaConf as a float
            return float(value)
        return value
    expr = m.group(1)
    if ":" in expr:  # custom resolver (oc.env, hydra, now): left as it is
        return value
    dots = len(expr) - len(expr.lstrip("."))
    keys = [k for k in expr.lstrip(".").split(".") if k != ""]
    base = list(holder[: len(holder) - (dots - 1)]) if dots else []
    target_path = base + keys
    target = _lookup(root, target_path)
    return _resolve_value(root, target_path[:-1], target, depth + 1)


def resolve(root: Dict[str, Any], path: List[Any]) -> Any:
    """The subtree at `path` with every interpolation replaced by its (deep-copied, recursively resolved) target."""
    node = _resolve_value(root, path[:-1], _lookup(root, path))
    # after following a whole-node interpolation the holder of the children is the TARGET's path: find it again
    here = path
    raw = _lookup(root, path)
    hops = 0
    while isinstance(raw, str) and _INTERP.fullmatch(raw.strip()) and ":" not in raw:
        expr = _INTERP.fullmatch(raw.strip()).group(1)
        dots = len(expr) - len(expr.lstrip("."))
        keys = [k for k in expr.lstrip(".").split(".") if k != ""]
        holder = here[:-1]
        here = (list(holder[: len(holder) - (dots - 1)]) if dots else []) + keys
        raw = _lookup(root, here)
        hops += 1
        if hops > 32:
            raise RecursionError("interpolation cycle")
    if isinstance(node, dict):
        return {k: resolve(root, here + [k]) for k in node}
    if isinstance(node, list):
        return [resolve(root, here + [i]) for i in range(len(node))]
    return node


def plugin_kwargs(experiment: str) -> Dict[str, Any]:
    """What `instantiate(self._backbone / _head / _decoder)` receives (``_recursive_: false``: nested configs stay configs), plus the
    two model-level entries the reference passes at call time (``tasks``, ``post_processing_config``: detector.py:352-362)."""
    root = compose(experiment)
    out = {k: resolve(root, ["model", k]) for k in ("_backbone", "_head", "_decoder", "post_processing_config", "tasks")}
    out["trainer"] = {k: resolve(root, ["trainer", k]) for k in ("precision", "sync_batchnorm", "gradient_clip_val")}
    out["batch_size"] = resolve(root, ["model", "batch_size"])
    out["range_view_config"] = resolve(root, ["dataset", "_train_dataset", "range_view_config"])
    return out


def swap_targets(kwargs: Dict[str, Any]) -> Dict[str, Any]:
    """the three top-level `_target_`s change, nothing else."""
    out = copy.deepcopy(kwargs)
    for k in ("_backbone", "_head", "_decoder"):
        out[k]["_target_"] = TARGET_SWAP[out[k]["_target_"]]
    return out


# ---- JSON fixture (data): integer keys and non-finite floats need a wire form -----------------------------------------------
def to_wire(x: Any) -> Any:
    if isinstance(x, dict):
        return {"__dict__": [[to_wire(k), to_wire(v)] for k, v in x.items()]}
    if isinstance(x, list):
        return [to_wire(v) for v in x]
    if isinstance(x, float) and not math.isfinite(x):
        return {"__float__": repr(x)}
    return x


def from_wire(x: Any) -> Any:
    if isinstance(x, dict):
        if "__float__" in x:
            return float(x["__float__"])
        return {from_wire(k): from_wire(v) for k, v in x["__dict__"]}
    if isinstance(x, list):
        return [from_wire(v) for v in x]
    return x


def load_fixture() -> Dict[str, Any]:
    with open(FIXTURE) as f:
        return from_wire(json.load(f))


if __name__ == "__main__":
    data = {e: plugin_kwargs(e) for e in ("rv-av2", "rv-waymo")}
    with open(FIXTURE, "w") as f:
        json.dump(to_wire(data), f, indent=1, sort_keys=False)
        f.write("\n")
    print("wrote", FIXTURE, os.path.getsize(FIXTURE), "bytes")
