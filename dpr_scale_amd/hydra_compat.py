"""The slice of Hydra 1.1 that dpr-scale's training path relies on, over PyYAML -- used only when `hydra` is
not importable (not installed in the build image, no network).  With hydra present the task uses the real
`hydra.utils.instantiate` and the recipes run through the real `@hydra.main`.

  instantiate(conf, *args, **kwargs)   `_target_` class path + kwargs; `_recursive_=False` leaves nested configs
                                       as configs (main.py:25 instantiates the task that way)
  compose(conf_dir, name, overrides)   defaults lists (incl. `override grp: opt`), `# @package _group_` /
                                       `_global_` headers, `${a.b}` interpolation, `k=v` / `+k=v` overrides
It loads the reference's own conf/ tree unchanged (tests/test_task_dropin.py does so when /root/reference is
mounted) as well as the small tree under dpr_scale_amd/conf.
"""
import copy
import importlib
import os
import re

import yaml


class Conf(dict):
    """dict with attribute access (enough of OmegaConf's DictConfig for the task and main)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_conf(x):
    if isinstance(x, dict):
        return Conf({k: to_conf(v) for k, v in x.items()})
    if isinstance(x, list):
        return [to_conf(v) for v in x]
    return x


def _locate(path):
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def instantiate(conf, *args, _recursive_=True, **kwargs):
    if conf is None:
        return None
    conf = dict(conf)
    conf.update(kwargs)
    target = conf.pop("_target_")
    recursive = conf.pop("_recursive_", _recursive_)
    cls = _locate(target) if isinstance(target, str) else target
    kw = {}
    for k, v in conf.items():
        if recursive and isinstance(v, dict) and "_target_" in v:
            v = instantiate(v, _recursive_=True)
        kw[k] = v
    return cls(*args, **kw)


# ---- config composition ------------------------------------------------------------------------------------
_SCI = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")


def _floats(node):
    """OmegaConf reads `3e-5` as a float; PyYAML (YAML 1.1) leaves it a string."""
    if isinstance(node, dict):
        return {k: _floats(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_floats(v) for v in node]
    if isinstance(node, str) and _SCI.match(node):
        return float(node)
    return node


def _read(conf_dir, rel):
    path = os.path.join(conf_dir, rel if rel.endswith(".yaml") else rel + ".yaml")
    text = open(path).read()
    m = re.search(r"#\s*@package\s+(\S+)", text)
    return _floats(yaml.safe_load(text) or {}), (m.group(1) if m else None)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(root, dotted, value):
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = value


def _get_path(root, dotted):
    cur = root
    for p in dotted.split("."):
        cur = cur[p]
    return cur


def _interpolate(node, root):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _interpolate(node[k], root)
    elif isinstance(node, list):
        return [_interpolate(v, root) for v in node]
    elif isinstance(node, str):
        m = re.fullmatch(r"\$\{([^}]+)\}", node)
        if m:
            return _interpolate(_get_path(root, m.group(1)), root)
        return re.sub(r"\$\{([^}]+)\}", lambda mm: str(_get_path(root, mm.group(1))), node)
    return node


# the reference's root defaults live in a Python dataclass (conf/config.py:9-24); same list, as data
ROOT_DEFAULTS = [{"task": "dpr"}, {"task/model": "hf_model"}, {"task/transform": "hf_transform"}, {"task/optim": "adamw"},
                 {"datamodule": "default"}, {"trainer": "gpu_1_host"}, {"checkpoint_callback": "default"}]


def compose(conf_dir, config_name="config", overrides=()):
    """Returns the composed config as a Conf.  `config_name` may be a recipe such as msmarco_baseline.yaml."""
    groups = {}
    order = []

    def add_defaults(dl):
        for item in dl:
            if isinstance(item, str):
                continue  # "_self_" / "config": handled by the caller
            for k, v in item.items():
                k = k.replace("override ", "").strip()
                if k not in groups:
                    order.append(k)
                groups[k] = v

    recipe = {}
    root_first = {}  # the root's own keys when its defaults list starts with _self_ (groups then override them)
    root_file = os.path.join(conf_dir, "config.yaml")
    if os.path.isfile(root_file):
        root_cfg, _ = _read(conf_dir, "config")
        dl = root_cfg.pop("defaults", [])
        add_defaults(dl)
        if dl and dl[0] == "_self_":
            root_first = root_cfg
        else:
            recipe = root_cfg
    else:
        add_defaults(ROOT_DEFAULTS)
        root_first = {"test_only": False}
    name = config_name[:-5] if config_name.endswith(".yaml") else config_name
    if name != "config":
        rc, _ = _read(conf_dir, name)
        add_defaults(rc.pop("defaults", []))
        recipe = _merge(recipe, rc)
    for ov in overrides:  # group choices first (task/optim=lamb)
        k, _, v = ov.lstrip("+").partition("=")
        if k in groups or os.path.isdir(os.path.join(conf_dir, k)):
            groups[k] = v
            if k not in order:
                order.append(k)
    cfg = copy.deepcopy(root_first)
    for g in order:
        body, package = _read(conf_dir, os.path.join(g, str(groups[g])))
        body.pop("defaults", None)
        if package == "_global_":
            _merge(cfg, body)
        else:
            node = cfg
            for p in g.split("/"):
                node = node.setdefault(p, {})
            _merge(node, body)
    _merge(cfg, recipe)
    for ov in overrides:
        k, _, v = ov.lstrip("+").partition("=")
        if k in groups:
            continue
        _set_path(cfg, k, _floats(yaml.safe_load(v)))
    return to_conf(_interpolate(cfg, cfg))
