"""MI355X-native hctr inference engine (package root).

The directory name contains hyphens, so import it with
``importlib.import_module("handwritten-chinese-ocr-samples_amd")`` or through the
``hctr_amd`` alias module at the repository root.
"""
