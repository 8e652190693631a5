"""Line preprocessing in front of the hot path, resize on the device.

Mirrors the reference's two loaders:
  * ``test.py:204-227``  ``preprocess_input(input, height)`` / ``read_resize_image``: cv2.imread -> BGR2GRAY ->
    ``tw = int(height * (w / h))`` -> ``cv2.resize(..., interpolation=cv2.INTER_AREA)``
  * ``utils/dataset.py:47-60``  ``ImageDataset.pil_loader``: ``new_width = int(width * (img_h / height))``
  * ``utils/dataset.py:111-148``  ``AlignCollate``: batch width = min(max line width, max_width); wider lines are
    cropped and their labels cut proportionally

Files are decoded on the host with PIL (cv2 is not a dependency); everything after the decode - gray conversion,
INTER_AREA resize to height 128, packing into the padded ``[B,128,maxW]`` uint8 batch - runs in one kernel
(csrc/preprocess.hip) and the batch can stay in HBM for ``hctr_model.greedy``. Pixel parity with cv2 is unpinned
(DESIGN.md section 2): the kernel is bit-exact against oracle/resize_ref.py, a restatement of OpenCV's published
algorithm.
"""
import ctypes
import os

import numpy as np

from . import _lib

IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp")


def target_width(h, w, height=128, rule="test"):
    """'test': test.py:211-213 (ratio = w / h; tw = int(height * ratio)); 'dataset': utils/dataset.py:54-56."""
    if rule == "test":
        return int(height * (float(w) / float(h)))
    if rule == "dataset":
        return int(w * (height / h))
    raise ValueError("rule must be 'test' or 'dataset'")


def load_image(path):
    """Decode one file to u8 [H,W] (8-bit gray) or u8 [H,W,3] RGB."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("L", "RGB"):
            im = im.convert("L" if im.mode in ("1", "I;16", "I", "F", "LA") else "RGB")
        return np.asarray(im, dtype=np.uint8)


def resize_lines(model, images, height=None, rule="test", order="rgb", max_width=None, device_out=False):
    """Resize decoded images to ``height`` on ``model``'s GPU.

    images: list of u8 arrays, [H,W] gray or [H,W,3] colour (``order`` 'rgb' as PIL decodes, 'bgr' as cv2 does).
    Returns (batch, widths): batch u8 [n,height,maxW] - a numpy array, or a torch CUDA tensor when
    ``device_out`` - and int32 widths[n] (after the optional AlignCollate ``max_width`` crop)."""
    ctx = model._require_ctx()
    height = int(height or model.img_height)
    if order not in ("rgb", "bgr"):
        raise ValueError("order must be 'rgb' or 'bgr'")
    n = len(images)
    if n == 0:
        return np.zeros((0, height, 0), np.uint8), np.zeros((0,), np.int32)
    hs, ws, chs, offs, tws, flat = [], [], [], [], [], []
    pos = 0
    for i, im in enumerate(images):
        a = np.ascontiguousarray(im)
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3)):
            raise ValueError("image %d: expected uint8 [H,W] or [H,W,3], got %s %s" % (i, a.dtype, a.shape))
        if a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError("image %d is empty" % i)
        h, w = a.shape[:2]
        ch = 1 if (a.ndim == 2 or a.shape[2] == 1) else (3 if order == "bgr" else -3)
        tw = target_width(h, w, height, rule)
        if tw < 1:                                         # cv2.resize raises on an empty dsize
            raise ValueError("image %d (%dx%d) resizes to width 0" % (i, w, h))
        hs.append(h); ws.append(w); chs.append(ch); offs.append(pos); tws.append(tw)
        flat.append(a.reshape(-1))
        pos += a.size
    packed = np.concatenate(flat)
    out_w = max(tws)
    if max_width and out_w > max_width:
        out_w = int(max_width)
    hs, ws, chs, tws = (np.asarray(v, np.int32) for v in (hs, ws, chs, tws))
    offs = np.asarray(offs, np.int64)
    lib = _lib.load()
    if device_out:
        import torch
        out = torch.empty((n, height, out_w), dtype=torch.uint8, device="cuda:%d" % model._device)
        optr = ctypes.c_void_p(out.data_ptr())
    else:
        out = np.empty((n, height, out_w), np.uint8)
        optr = _lib.ptr(out)
    _lib.check(lib.hctr_resize_lines(ctx, _lib.ptr(packed), packed.size, offs.ctypes.data_as(_lib.c_i64p),
                                     hs.ctypes.data_as(_lib.c_i32p), ws.ctypes.data_as(_lib.c_i32p),
                                     chs.ctypes.data_as(_lib.c_i32p), n, height, tws.ctypes.data_as(_lib.c_i32p),
                                     out_w, optr, int(device_out)), ctx)
    return out, np.minimum(tws, out_w).astype(np.int32)


def list_inputs(path):
    """A file, or the image files of a folder (test.py:218-226; sorted here, os.listdir order there)."""
    if os.path.isfile(path):
        return [path]
    return [os.path.join(path, n) for n in sorted(os.listdir(path)) if n.lower().endswith(IMG_EXT)]


def preprocess_input(model, input, height=None, batch_size=256):
    """test.py:204-227: every image of ``input`` (file or folder) resized to ``height``; list of u8 [height,tw]."""
    paths = list_inputs(input)
    out = []
    for i in range(0, len(paths), batch_size):
        batch, widths = resize_lines(model, [load_image(p) for p in paths[i:i + batch_size]], height, "test")
        out += [batch[j, :, :widths[j]].copy() for j in range(len(widths))]
    return out


def truncate_label(label, w, max_w):
    """utils/dataset.py:139-143: a line cropped to the width cap keeps a proportional prefix of its label."""
    if w > max_w:
        return label[:max(1, int(len(label) * (max_w / w)))]
    return label
