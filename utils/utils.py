"""Config-to-object plumbing at the drop-in boundary.

MuDG's YAML configs name classes by dotted path (`target:`) with constructor kwargs (`params:`); the reference
resolves them in utils/utils.py:27-42.  Same contract here, so the reference's configs instantiate this package's
`lvdm.*` classes unchanged.  (The reference module also holds cv2 video helpers; they are host I/O and out of scope.)
"""
import importlib


def get_obj_from_str(string, reload=False):
    module_name, _, attr = string.rpartition(".")
    module = importlib.import_module(module_name)
    if reload:
        module = importlib.reload(module)
    return getattr(module, attr)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", None)
    return get_obj_from_str(config["target"])(**(dict(params) if params is not None else {}))


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def check_istarget(name, para_list):
    """True when `name` contains any entry of para_list (reference utils/utils.py:15-24: selects trainable parameters)."""
    return any(para in name for para in para_list)
