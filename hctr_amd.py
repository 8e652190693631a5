"""Importable alias for the hyphenated package directory ``handwritten-chinese-ocr-samples_amd``.

    from hctr_amd import hctr_model, ctc_codec
"""
import importlib as _importlib
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.abspath(__file__))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_pkg = _importlib.import_module("handwritten-chinese-ocr-samples_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
package = _pkg
__all__ = list(_pkg.__all__) + ["package"]
