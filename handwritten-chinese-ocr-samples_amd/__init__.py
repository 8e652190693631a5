"""MI355X-native hctr inference engine (package root).

The directory name contains hyphens (it mirrors the reference repository's name), so import it with
``importlib.import_module("handwritten-chinese-ocr-samples_amd")`` or through the ``hctr_amd`` alias
module at the repository root.

Public surface = the reference's surface for this path:
  hctr_model   drop-in for models/handwritten_ctr_model.py:156 (engine-backed)
  ctc_codec    drop-in for utils/ctc_codec.py:14                (engine-backed)
plus ``synth`` (deterministic synthetic checkpoints / line images) and ``build`` / ``load_library``.
"""
from . import preprocess, synth  # noqa: F401
from ._lib import build, load as load_library  # noqa: F401
from .codec import ArpaLM, ToyBigramLM, ZeroLM, ctc_codec  # noqa: F401
from .model import hctr_model  # noqa: F401

__all__ = ["hctr_model", "ctc_codec", "ZeroLM", "ToyBigramLM", "ArpaLM", "synth", "preprocess", "build", "load_library"]
