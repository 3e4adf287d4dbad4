"""ORACLE SCAFFOLDING (tests / golden generation only; never imported by the product).

Imports the REAL reference modules from /root/reference with `sys.modules` shims for the third-party
packages that are not installed here (timm, clip, einops_exts, cv2) and a patch for two dead imports
in `models/gpt2.py:14` that transformers 5.x removed.  /root/reference only exists
in the build container, so everything here is gated on `available()`; the GPU box uses the committed
fixtures in tests/golden/ and the restatement in oracle/torch_ref.py instead.
"""
import importlib
import os
import sys

REF_ROOT = os.environ.get("DVLA_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


class _RefImport:
    """Context manager: put shims + reference first on sys.path and hide this repo's own `models`
    drop-in package so `import models.*` resolves to the reference."""

    def __enter__(self):
        self._saved_path = list(sys.path)
        self._saved_mods = {k: v for k, v in sys.modules.items()
                            if k == "models" or k.startswith("models.") or k == "utils" or k.startswith("utils.")}
        for k in self._saved_mods:
            del sys.modules[k]
        # the reference's `models/` has no __init__.py (namespace package): a regular `models` package anywhere later
        # on sys.path -- this repo's drop-in package -- would win, so hide the repo root while importing.
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != repo]
        sys.path[:0] = [_SHIMS, REF_ROOT]
        self._saved_dwb = sys.dont_write_bytecode
        sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
        import transformers.pytorch_utils as pu
        for name in ("find_pruneable_heads_and_indices", "prune_conv1d_layer"):
            if not hasattr(pu, name):
                setattr(pu, name, lambda *a, **k: (_ for _ in ()).throw(NotImplementedError(name)))
        return self

    def __exit__(self, *exc):
        ref_mods = {k: v for k, v in sys.modules.items()
                    if k == "models" or k.startswith("models.") or k == "utils" or k.startswith("utils.")}
        for k in ref_mods:
            del sys.modules[k]
        sys.modules.update(self._saved_mods)
        sys.path[:] = self._saved_path
        sys.dont_write_bytecode = self._saved_dwb
        return False


_cache = {}


def ref_module(name):
    """Return reference module `name` (e.g. 'models.gpt2'); cached, isolated from the repo's `models`."""
    if not available():
        raise RuntimeError("reference tree not present (expected at %s)" % REF_ROOT)
    if name not in _cache:
        with _RefImport():
            _cache[name] = importlib.import_module(name)
            # keep sub-imports alive in the cache too
            for k, v in list(sys.modules.items()):
                if (k.startswith("models") or k.startswith("utils")) and k not in _cache:
                    _cache[k] = v
    return _cache[name]
