"""The reference's Hydra / OmegaConf task configs read as plain YAML (hydra and omegaconf are not dependencies of this repository).

`compose(task_name, overrides)` does what `hydra.compose(config_name="config", overrides=["task=<name>"])` does for the files the
reference ships (reference __init__.py:36-41): a root config, `cfg.task` from cfg/task/<name>.yaml and — when present —
`cfg.train` from cfg/train/<name>PPO.yaml, with the OmegaConf interpolations those files use resolved:

    ${..physics_engine}  ${...num_envs}  ${.name}  ${a.b.c}          relative (one dot = the node the value sits in) and absolute
    ${resolve_default:4096,${...num_envs}}  ${eq:a,b}  ${contains:a,b}  ${if:p,a,b}      the four resolvers of reference __init__.py:8-11

The reference has NO root cfg/config.yaml: `ROOT_DEFAULTS` restates the keys its task / train yamls interpolate, with
upstream IsaacGymEnvs' defaults for the standard ones and '' (= "use the yaml's own default", the resolve_default convention)
for the fork-specific reward overrides.
"""
import copy
import os

import yaml

ROOT_DEFAULTS = dict(
    task_name="", experiment="", num_envs="", seed=42, torch_deterministic=False, max_iterations="", physics_engine="physx", pipeline="gpu",
    sim_device="cuda:0", rl_device="cuda:0", graphics_device_id=0, num_threads=4, solver_type=1, num_subscenes=4, test=False, checkpoint="",
    sigma="", multi_gpu=False, wandb_activate=False, wandb_name="", capture_video=False, capture_video_freq=1464, capture_video_len=100,
    force_render=True, headless=True, pbt=dict(enabled=False),
    # fork-specific command-line overrides the task yamls interpolate ('' = keep the yaml's default)
    alpha_velocity_reward="", power_coefficient="", penalty="", hit_reward="", hit_penalty="", cross_net_reward="", die_penalty="",
    hit_paddle_reward="", miss_paddle_penalty_coefficient="",
)

RESOLVERS = {   # reference __init__.py:8-11
    "eq": lambda x, y: str(x).lower() == str(y).lower(),
    "contains": lambda x, y: str(x).lower() in str(y).lower(),
    "if": lambda pred, a, b: a if pred else b,
    "resolve_default": lambda default, arg: default if arg == "" else arg,
}


class InterpolationError(KeyError):
    pass


def _literal(text):
    """An unquoted / quoted resolver argument as a Python value ('4096' -> 4096, '"gpu"' -> 'gpu', '' -> '')."""
    t = text.strip()
    if t == "":
        return ""
    try:
        return yaml.safe_load(t)
    except yaml.YAMLError:
        return t


def _split_top(s, sep):
    """Split on `sep` outside ${...} nesting and quotes."""
    out, depth, cur, quote, i = [], 0, [], None, 0
    while i < len(s):
        c = s[i]
        if quote:
            cur.append(c)
            if c == quote:
                quote = None
        elif c in "\"'":
            quote = c
            cur.append(c)
        elif s.startswith("${", i):
            depth += 1
            cur.append("${")
            i += 1
        elif c == "}" and depth > 0:
            depth -= 1
            cur.append(c)
        elif c == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
        i += 1
    out.append("".join(cur))
    return out


def _lookup(root, path, ref):
    """`ref` without the braces: '..a.b' (relative: one dot = the container of the value at `path`) or 'a.b' (absolute)."""
    dots = len(ref) - len(ref.lstrip("."))
    keys = [k for k in ref[dots:].split(".") if k != ""]
    if dots:
        base = list(path[:-1])                 # the container the value sits in
        if dots - 1 > len(base):
            raise InterpolationError(f"'${{{ref}}}' at {'.'.join(map(str, path))} climbs above the root")
        base = base[: len(base) - (dots - 1)]
    else:
        base = []
    node, where = root, base + keys
    for k in where:
        if isinstance(node, dict) and k in node:
            node = node[k]
        elif isinstance(node, list) and str(k).isdigit() and int(k) < len(node):
            node = node[int(k)]
        else:
            raise InterpolationError(f"'${{{ref}}}' at {'.'.join(map(str, path))}: no key {'.'.join(map(str, where))}")
    return node, where


def _resolve_str(root, path, s, stack):
    """Resolve every ${...} in s.  A string that IS one interpolation keeps the referenced value's type."""
    if "${" not in s:
        return s
    pieces, i, whole = [], 0, None
    while i < len(s):
        j = s.find("${", i)
        if j < 0:
            pieces.append(s[i:])
            break
        pieces.append(s[i:j])
        depth, k = 0, j
        while k < len(s):
            if s.startswith("${", k):
                depth += 1
                k += 2
                continue
            if s[k] == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        if depth != 0:
            raise InterpolationError(f"unbalanced interpolation in {s!r}")
        inner = s[j + 2:k]
        head = _split_top(inner, ":")
        if len(head) > 1 and head[0].strip() in RESOLVERS:
            args = [_resolve_str(root, path, a.strip(), stack) if "${" in a else _literal(a) for a in _split_top(":".join(head[1:]), ",")]
            val = RESOLVERS[head[0].strip()](*args)
        else:
            node, where = _lookup(root, path, inner.strip())
            key = tuple(where)
            if key in stack:
                raise InterpolationError(f"interpolation cycle through {'.'.join(map(str, where))}")
            val = _resolve_node(root, where, node, stack | {key})
        if j == 0 and k == len(s) - 1:
            whole = val
        pieces.append(val)
        i = k + 1
    if whole is not None or (len(pieces) == 2 and pieces[0] == ""):
        return pieces[1] if whole is None else whole
    return "".join(str(p) for p in pieces)


def _resolve_node(root, path, node, stack=frozenset()):
    if isinstance(node, dict):
        return {k: _resolve_node(root, list(path) + [k], v, stack) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve_node(root, list(path) + [i], v, stack) for i, v in enumerate(node)]
    if isinstance(node, str):
        return _resolve_str(root, list(path), node, stack)
    return node


def resolve(root):
    """A deep copy of `root` (nested dicts / lists) with every interpolation resolved."""
    root = copy.deepcopy(root)
    return _resolve_node(root, [], root)


def compose(task_name, cfg_dir, overrides=None, with_train=True):
    """Root config + cfg/task/<task_name>.yaml (+ cfg/train/<task_name>PPO.yaml), interpolations resolved.  `overrides`: root keys."""
    root = copy.deepcopy(ROOT_DEFAULTS)
    root.update(overrides or {})
    root["task_name"] = task_name
    with open(os.path.join(cfg_dir, "task", f"{task_name}.yaml")) as fh:
        root["task"] = yaml.safe_load(fh)
    train = os.path.join(cfg_dir, "train", f"{task_name}PPO.yaml")
    if with_train and os.path.exists(train):
        with open(train) as fh:
            root["train"] = yaml.safe_load(fh)
    return resolve(root)
