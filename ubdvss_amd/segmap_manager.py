"""SegmapManager.postprocess -- host mirror of semantic_segmentation/segmap_manager.py:41-69.

Same static-method signature as the reference; the work (external components, contourArea
filter, minAreaRect, boxPoints, class vote) runs in libubd_hip.so on the MI355X.
"""
import ctypes
import logging

import numpy as np
import torch
from PIL import Image, ImageDraw

from . import _lib
from .data_markup import ObjectMarkup, ClassifiedObjectMarkup

_handles = {}
_workspaces = _lib.StreamWorkspaces()   # (device, stream) -> uint8 tensor: the static entry points reuse their scratch between calls (LRU-bounded)


def _workspace(device, nbytes):
    """Device scratch of at least nbytes, kept per (device, current stream): calls on one stream are ordered, two streams
    never share a scratch (grown when a bigger request comes; never shrunk)."""
    return _workspaces.get(device, torch.cuda.current_stream(device).cuda_stream, nbytes)


def _reset_handles():
    """Destroys the cached C-ABI handles (they read the UBD_* test switches when they are created)."""
    lib = _lib.load()
    for h in _handles.values():
        lib.ubd_destroy(h)
    _handles.clear()


def _handle(n_classes, device):
    key = (n_classes, str(device))
    if key not in _handles:
        lib = _lib.load()
        cfg = _lib.UbdConfig(1, n_classes, 1, _lib.UBD_F32)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.ubd_create(ctypes.byref(cfg), ctypes.byref(h)), "ubd_create")
        _handles[key] = h
    return _handles[key]


class SegmapManager:
    @staticmethod
    def postprocess(seg_map, seg_map_class_logits=None, scale=1, min_area_threshold=5, max_objects=1024):
        """seg_map: (h,w,1) or (h,w) array of {0,1}; seg_map_class_logits: (h,w,n_cls) or None.
        Returns list[ObjectMarkup] / list[ClassifiedObjectMarkup] like the reference."""
        if not torch.cuda.is_available():
            raise RuntimeError("SegmapManager.postprocess needs an MI355X; there is no CPU fallback")
        lib = _lib.load()
        device = torch.device(f"cuda:{torch.cuda.current_device()}")
        m = np.asarray(seg_map)
        m = m.reshape(m.shape[0], m.shape[1])
        h, w = m.shape
        n_cls = 0 if seg_map_class_logits is None else int(np.shape(seg_map_class_logits)[-1])
        # pack (map, class logits) as a (1,h,w,1+n_cls) logits tensor; map != 0 <=> "logit" 1 > 0.5
        lg = np.zeros((1, h, w, 1 + n_cls), np.float32)
        lg[0, :, :, 0] = (m != 0)
        if n_cls:
            lg[0, :, :, 1:] = np.asarray(seg_map_class_logits, np.float32)
        lgt = torch.from_numpy(lg).to(device)
        hd = _handle(n_cls, device)
        cap = max_objects
        quads = torch.empty((1, cap, 8), dtype=torch.int32, device=device)      # the library writes the count and the list's first entries
        classes = torch.empty((1, cap), dtype=torch.int32, device=device)
        counts = torch.empty((1,), dtype=torch.int32, device=device)
        ws = _workspace(device, lib.ubd_postprocess_workspace_bytes(hd, 1, h, w, cap))
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _lib.check(lib.ubd_postprocess(hd, lgt.data_ptr(), 1, h, w, 0.5, int(scale), float(min_area_threshold),
                                       None, quads.data_ptr(), classes.data_ptr(), counts.data_ptr(), cap,
                                       ws.data_ptr(), ws.numel(), stream), "ubd_postprocess")
        n = int(counts.cpu()[0])
        if n > cap:
            raise RuntimeError(f"more than max_objects={cap} objects found ({n})")
        q = quads.cpu().numpy()[0, :n].astype(int)
        if seg_map_class_logits is None:
            return [ObjectMarkup(bbox) for bbox in q]
        c = classes.cpu().numpy()[0, :n]
        return [ClassifiedObjectMarkup(bbox, class_id) for bbox, class_id in zip(q, c)]

    @staticmethod
    def prepare_image_and_target(image, markup, net_config, augment=False):
        """segmap_manager.py:24-39 without the augmentation branch (imgaug pipeline: outside the hot path):
        rescale image + markup to the network's size rule, build the label map at ``net_config.get_scale()``.
        Returns (image, markup, label map) like the reference."""
        if augment:
            raise ValueError("augmentation is not part of this package (SURVEY.md section 2: out of scope); augment the "
                             "image and markup first, then call prepare_image_and_target(..., augment=False)")
        image, markup = SegmapManager._rescale_image_and_markup(image, markup, net_config)
        return image, markup, SegmapManager.build_segmentation_map(image, markup, scale=net_config.get_scale())

    @staticmethod
    def build_segmentation_map(image, markup, scale=1, for_drawing=False):
        """Training label map (behaviour of segmap_manager.py:81-104): every quad is divided by ``scale``, its
        corners are snapped outward (``_proper_round``) and the polygon is filled, later objects over earlier
        ones, with 255 (``for_drawing``), class id + 1 (classified markup) or 1.  Returns a PIL 'L' image of
        size (W/scale, H/scale)."""
        width, height = image.size
        if width % scale or height % scale:
            raise AssertionError("image size must be a multiple of the map scale")
        canvas = Image.new('L', (width // scale, height // scale), 0)
        pen = ImageDraw.Draw(canvas)
        for obj in markup:
            value = 255 if for_drawing else (getattr(obj, "object_type", 0) + 1 if isinstance(obj, ClassifiedObjectMarkup) else 1)
            if value > 255:
                raise AssertionError("No more than 255 classes are supported")
            corners = SegmapManager._proper_round(np.asarray(obj.bbox, dtype=np.float64) / scale)
            pen.polygon([int(v) for v in corners], fill=int(value))
        return canvas

    @staticmethod
    def build_segmentation_maps_on_device(image_size, markups, scale=1, for_drawing=False, device=None, strict_markup=False):
        """Batch form of ``build_segmentation_map`` on the MI355X (ubd_build_label_maps): ``markups`` is a list (one entry
        per image) of lists of ObjectMarkup / ClassifiedObjectMarkup, ``image_size`` = (width, height) like ``PIL.Image.size``.
        Returns an int32 device tensor (N, height/scale, width/scale) -- the y_true layout of the loss / train step.
        Degenerate quads whose OPPOSITE corners coincide on the map (a point or a folded segment -- a labelling error) are drawn
        like every other quad and logged; ``strict_markup=True`` refuses them instead (see ``_check_folded_quads``)."""
        if not torch.cuda.is_available():
            raise RuntimeError("SegmapManager.build_segmentation_maps_on_device needs an MI355X; there is no CPU fallback")
        lib = _lib.load()
        device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        width, height = image_size
        if width % scale or height % scale:
            raise AssertionError("image size must be a multiple of the map scale")
        n = len(markups)
        cap = max(1, max((len(m) for m in markups), default=1))
        # float64 markup, as it reaches the reference's `object_markup.bbox / scale` (segmap_manager.py:96): rescaled or
        # augmented quads are fractional, and _proper_round floors / ceils the QUOTIENT -- truncating the markup first
        # would move a corner by a whole map pixel (12.3 / 4 ceils to 4, 12 / 4 to 3)
        quads = np.zeros((n, cap, 8), np.float64)
        values = np.zeros((n, cap), np.int32)
        counts = np.zeros((n,), np.int32)
        for i, objs in enumerate(markups):
            counts[i] = len(objs)
            for j, obj in enumerate(objs):
                value = 255 if for_drawing else (getattr(obj, "object_type", 0) + 1 if isinstance(obj, ClassifiedObjectMarkup) else 1)
                if value > 255:
                    raise AssertionError("No more than 255 classes are supported")
                bbox = np.asarray(obj.bbox, dtype=np.float64).reshape(-1)
                if bbox.size != 8:
                    raise ValueError(f"object markup must be a quadrilateral (8 numbers), got {bbox.size} (image {i}, object {j})")
                quads[i, j] = bbox
                values[i, j] = value
        SegmapManager._check_folded_quads(quads, counts, scale, strict_markup)
        qd, vd, cd = (torch.from_numpy(a).to(device) for a in (quads, values, counts))
        labels = torch.empty((n, height // scale, width // scale), dtype=torch.int32, device=device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        with torch.cuda.device(device):
            _lib.check(lib.ubd_build_label_maps(qd.data_ptr(), vd.data_ptr(), cd.data_ptr(), n, cap, height // scale, width // scale,
                                                int(scale), labels.data_ptr(), stream), "ubd_build_label_maps")
        return labels

    @staticmethod
    def _check_folded_quads(quads, counts, scale, strict=False):
        """The device fill rule is Pillow's for every quadrilateral except a zero-area fold whose OPPOSITE corners coincide
        after snapping (four edges in one point: Pillow's corner joining there depends on its internal edge order and is not
        restated, oracle/label_raster.py).  The reference draws whatever ``ImageDraw.polygon`` draws for such markup
        (segmap_manager.py:93-103) and carries on; so does this builder -- the device rule gives Pillow's pixels for ~96 % of random such
        quads and differs inside ONE row (a fragment of that row's span: median 2 pixels, 90th percentile 10) for the rest
        (tests/test_gpu_raster.py measures both) -- and logs
        a warning, because such an object is a labelling error.  ``strict``: raise instead.  Returns the number found."""
        pts = (quads / float(scale)).reshape(quads.shape[0], quads.shape[1], 4, 2)
        n_larger = (pts[:, :, None, :, :] > pts[:, :, :, None, :]).sum(axis=3)
        snapped = np.where(n_larger > 1, np.floor(pts), np.ceil(pts))
        folded = ((snapped[:, :, 0] == snapped[:, :, 2]).all(-1) | (snapped[:, :, 1] == snapped[:, :, 3]).all(-1))
        folded &= np.arange(quads.shape[1])[None, :] < np.asarray(counts)[:, None]
        if folded.any():
            i, j = np.argwhere(folded)[0]
            msg = (f"{int(folded.sum())} object(s) whose opposite corners coincide on the label map, first: object {j} of image {i} "
                   f"({quads[i, j].tolist()} at scale {scale})")
            if strict:
                raise ValueError(msg + "; fix the markup")
            logging.warning("label maps: %s -- drawn as a degenerate polygon (may differ from Pillow inside the fold point's row)", msg)
        return int(folded.sum())

    @staticmethod
    def _rescale_image_and_markup(image, markup, net_config, max_side=None):
        """Image and markup at the size the network wants (behaviour of segmap_manager.py:135-173): if the longer side exceeds
        ``max_side`` (default ``net_config.get_max_side()``) it becomes exactly ``max_side`` and the other side is scaled
        in proportion and rounded to a multiple of ``net_config.get_side_multiple()``; otherwise both sides are rounded to
        that multiple.  Rounding is Python 3's ``round`` (half to even), at least one multiple.  The image is resampled
        with ``Image.BICUBIC``; every quad is multiplied by (new_w / w, new_h / h) and rewrapped with
        ``create_same_markup`` (so it stays fractional, float64).  Empty / None markup is returned as it is.  Host-side
        by nature (a PIL image in, a PIL image out), like the reference."""
        w, h = image.size
        multiple = net_config.get_side_multiple()
        if max_side is None:
            max_side = net_config.get_max_side()

        def to_multiple(side):
            return max(1, round(side / multiple)) * multiple

        if max(w, h) > max_side:
            shrink = max_side / max(h, w)
            new_w, new_h = (max_side, to_multiple(h * shrink)) if w > h else (to_multiple(w * shrink), max_side)
        else:
            new_w, new_h = to_multiple(w), to_multiple(h)
        resized = image.resize(size=(new_w, new_h), resample=Image.BICUBIC)
        if not markup:
            return resized, markup
        factors = np.array([[new_w / w, new_h / h]])
        return resized, [m.create_same_markup((np.array(m.bbox).reshape((-1, 2)) * factors).reshape((-1,))) for m in markup]

    @staticmethod
    def _proper_round(markup_bbox):
        """Outward snapping of a quad's corners (behaviour of segmap_manager.py:106-133): a coordinate is floored
        when at least two of the four coordinates on the same axis are strictly larger (the object extends to
        the larger side), otherwise it is ceiled; anything that is not 4 points is truncated to int32."""
        pts = np.asarray(markup_bbox, dtype=np.float64)
        if pts.size != 8:
            return np.asarray(markup_bbox).astype(np.int32)
        pts = pts.reshape(4, 2)
        n_larger = (pts[None, :, :] > pts[:, None, :]).sum(axis=1)          # per corner, per axis
        snapped = np.where(n_larger > 1, np.floor(pts), np.ceil(pts))
        return snapped.reshape(-1).astype(np.int32)
