"""`lvdm.modules.encoders.condition` at the drop-in boundary — an import shim, not an implementation.

MuDG's YAML configs name the OpenCLIP towers by this path (configs/stage2-1024_mdm_waymo_infer.yaml:82,88 ->
reference lvdm/modules/encoders/condition.py: FrozenOpenCLIPEmbedder :174, FrozenOpenCLIPImageEmbedderV2 :295).  The
towers are third-party models with downloaded weights and are OUTSIDE the denoising path (SURVEY §8 / DESIGN §8): this
repo ships no CLIP code.  When this package's `lvdm/` is the one on the import path, the names below resolve lazily to
the classes of an externally provided condition module:

  * MUDG_CONDITION_MODULE=<dotted.module>   an importable module that defines them, or
  * MUDG_REFERENCE=<path to a MuDG checkout>  its lvdm/modules/encoders/condition.py is loaded under a private module
    name (it imports `lvdm.common.autocast` and `utils.utils.count_params`, which this overlay provides).

Without either, instantiating a tower raises with this explanation; importing this module and resolving the `target:`
strings never fails.  The towers' outputs enter the path as tensors (cond_stage_model.encode -> (B, 77, 1024),
embedder -> (B, 257, 1280) -> Resampler)."""
import importlib
import importlib.util
import os

_NAMES = ("AbstractEncoder", "IdentityEncoder", "ClassEmbedder", "FrozenT5Embedder", "FrozenCLIPEmbedder",
          "ClipImageEmbedder", "FrozenOpenCLIPEmbedder", "FrozenOpenCLIPImageEmbedder", "FrozenOpenCLIPImageEmbedderV2",
          "FrozenCLIPT5Encoder")
_external = None


def _load_external():
    global _external
    if _external is not None:
        return _external
    name = os.environ.get("MUDG_CONDITION_MODULE")
    if name:
        _external = importlib.import_module(name)
        return _external
    root = os.environ.get("MUDG_REFERENCE")
    if root:
        path = os.path.join(root, "lvdm", "modules", "encoders", "condition.py")
        if not os.path.isfile(path):
            raise ImportError(f"MUDG_REFERENCE={root!r}: {path} does not exist")
        spec = importlib.util.spec_from_file_location("_mudg_external_condition", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)          # needs open_clip / kornia / transformers, like the reference itself
        _external = mod
        return _external
    raise ImportError(
        "lvdm.modules.encoders.condition is an import shim: the OpenCLIP text / image towers are outside the MI355X "
        "denoising path and are not shipped here. Set MUDG_CONDITION_MODULE to an importable module that defines them, or "
        "MUDG_REFERENCE to a MuDG checkout whose lvdm/modules/encoders/condition.py should be used.")


class _Lazy:
    """Stands in for one tower class: `target:` resolution (getattr on this module) succeeds; construction resolves the
    real class from the external module."""

    def __init__(self, name):
        self.__name__ = self.__qualname__ = name

    def _real(self):
        return getattr(_load_external(), self.__name__)

    def __call__(self, *args, **kwargs):
        return self._real()(*args, **kwargs)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return getattr(self._real(), item)

    def __repr__(self):
        return f"<lazy {self.__name__} (resolved from MUDG_CONDITION_MODULE / MUDG_REFERENCE on use)>"


for _n in _NAMES:
    globals()[_n] = _Lazy(_n)
del _n
