"""isaacgymenvs.utils.reformat (train.py:87): `omegaconf_to_dict`, `print_dict`.  Works on OmegaConf nodes when omegaconf is
installed and on plain dicts / attribute dicts otherwise (this repository depends on neither hydra nor omegaconf)."""


def omegaconf_to_dict(d):
    """Nested DictConfig -> nested dict with interpolations resolved (upstream semantics); plain containers are deep-copied."""
    try:   # pragma: no cover - omegaconf is absent offline
        from omegaconf import DictConfig, ListConfig, OmegaConf
        if isinstance(d, (DictConfig, ListConfig)):
            return OmegaConf.to_container(d, resolve=True)
    except ImportError:
        pass
    if hasattr(d, "items"):
        return {k: omegaconf_to_dict(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [omegaconf_to_dict(v) for v in d]
    return d


def print_dict(val, nesting=-4, start=True):
    """Pretty-print a nested dict, upstream's layout."""
    if isinstance(val, dict):
        if not start:
            print("")
        nesting += 4
        for k in val:
            print(nesting * " ", end="")
            print(k, end=": ")
            print_dict(val[k], nesting, start=False)
    else:
        print(val)
