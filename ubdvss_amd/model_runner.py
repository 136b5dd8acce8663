"""ModelRunner.predict -- host mirror of semantic_segmentation/model_runner.py:31-38, :105-148.

The reference runs ``model.predict`` on the device, then thresholds with numpy and calls
``SegmapManager.postprocess`` (OpenCV) image by image on the host.  Here the forward pass, the
threshold, the component labelling and the box fitting all run on the MI355X; only the final
(count, quads, classes) lists cross PCIe.
"""
import numpy as np
import torch

from . import _lib, utils
from .data_markup import ObjectMarkup, ClassifiedObjectMarkup


class ModelRunner:
    def __init__(self, net_config, pixel_threshold=0.5, max_objects_per_image=256, pipelined=False):
        """model_runner.py:31-38: pixel_probability > pixel_threshold is positive.
        pipelined=True: the postprocess of batch k is enqueued together with the forward pass of batch k+1 -- inside the
        first blocks of that pass's stem kernel (ubd_forward_postprocess), on the same stream, without events.  The result
        tensors a call returns are therefore COMPLETE only after the next call (or ``flush()``) has been enqueued and the
        stream has been waited for; ``synchronize()`` does both.  Logits / results are double-buffered: they stay valid until
        the second call after the one that returned them."""
        self._net_config = net_config
        eps = 1e-9
        self._logit_threshold = - np.log(1 / np.clip(pixel_threshold, eps, 1 - eps) - 1)
        self._cap = max_objects_per_image
        self._pipelined = pipelined
        self._slots = {}
        self._step = 0
        self._pending = None            # (model, slot): logits whose postprocess has not been enqueued yet

    def _job(self, slot):
        return {"logits": slot["logits"], "logit_threshold": self._logit_threshold, "scale": self._net_config.get_scale(),
                "min_area": self._net_config.get_min_pixels_for_detection(), "cap": self._cap, "outputs": slot["out"]}

    def flush(self):
        """Enqueues the postprocess that is still owed (the last batch's), as a launch of its own on the current stream."""
        if self._pending is not None:
            model, slot = self._pending
            self._pending = None
            job = self._job(slot)
            model.postprocess_on_device(job["logits"], job["logit_threshold"], job["scale"], job["min_area"], cap=job["cap"],
                                        outputs=job["outputs"])

    def synchronize(self):
        """All results returned so far are complete after this."""
        model = self._pending[0] if self._pending is not None else None
        self.flush()
        torch.cuda.current_stream(model.device if model is not None else None).synchronize()

    @property
    def logit_threshold(self):
        return self._logit_threshold

    def predict_on_device(self, model, images):
        """images: device tensor (N,H,W,C).  Returns device tensors
        (logits, binary_map (N,h,w) int32, quads (N,cap,8), classes or None, counts (N))."""
        scale = self._net_config.get_scale()
        min_area = self._net_config.get_min_pixels_for_detection()
        if not self._pipelined:
            logits = model.predict_on_device(images)
            bmap, quads, classes, counts = model.postprocess_on_device(logits, self._logit_threshold, scale, min_area, cap=self._cap)
            return logits, bmap, quads, classes, counts
        # ---- pipeline on ONE stream: this call = forward(batch k+1) + postprocess(batch k) in the same launches
        n, hh, ww, _ = images.shape
        key = (self._step & 1, n, hh, ww, id(model))
        self._step += 1
        slot = self._slots.get(key)
        if slot is None:
            slot = {"logits": torch.empty((n, hh // 4, ww // 4, model.k_out), dtype=torch.float32, device=model.device),
                    "out": model.alloc_postprocess_outputs(n, hh // 4, ww // 4, self._cap)}
            self._slots[key] = slot
        if self._pending is not None and (self._pending[0] is not model or self._pending[1] is slot):
            self.flush()                                # another model's logits (its own handle), or the slot about to be overwritten
        job = self._job(self._pending[1]) if self._pending is not None else None
        logits = model.predict_on_device(images, out=slot["logits"], postprocess=job)
        self._pending = (model, slot)
        bmap, quads, classes, counts = slot["out"]
        return logits, bmap, quads, classes, counts

    def predict(self, model, images, rescale=False, meta_infos=None):
        """Same contract as the reference (model_runner.py:105-138): returns
        (detection map (N,h,w,1) of {0,1}, classification_logits (N,h,w,n_cls), found_objects)."""
        assert not rescale or (meta_infos is not None and len(images) == len(meta_infos))
        x = np.asarray(images)
        if x.dtype != np.uint8:
            x = x.astype(np.float32, copy=False)        # no second 100-MB host copy when the batch is float32 already
        # float images arrive already preprocessed, as in the reference; uint8 images are raw pixels and take the
        # NetConfig preprocessing fused into the first layer (one rule for every entry point)
        xt = torch.from_numpy(np.ascontiguousarray(x)).to(model.device)
        logits, bmap, quads, classes, counts = self.predict_on_device(model, xt)
        self.flush()                                    # pipelined runner: this batch's postprocess now, the copies below wait for it
        counts_h = counts.cpu().numpy()
        if (counts_h > self._cap).any():
            raise RuntimeError(f"more than max_objects_per_image={self._cap} objects in an image "
                               f"(max found {int(counts_h.max())}); raise the capacity")
        quads_h = quads.cpu().numpy()
        classes_h = classes.cpu().numpy() if classes is not None else None
        logits_h = logits.cpu().numpy()
        detection_logits = bmap.cpu().numpy().astype(np.int64)[..., None]
        classification_logits = logits_h[..., 1:]
        with_cls = self._net_config.is_classification_supported()
        found_objects = []
        for i in range(x.shape[0]):
            objs = []
            for j in range(int(counts_h[i])):
                bbox = quads_h[i, j].astype(int)
                objs.append(ClassifiedObjectMarkup(bbox, classes_h[i, j]) if with_cls else ObjectMarkup(bbox))
            found_objects.append(objs)
        if rescale:
            found_objects = self.rescale(found_objects, meta_infos)
        return detection_logits, classification_logits, found_objects

    @staticmethod
    def rescale(found_objects, meta_infos):
        """Boxes back to the coordinates of the original images (behaviour of model_runner.py:140-148): per image,
        multiply x by meta.xscale and y by meta.yscale and truncate toward zero (utils.rescale_bbox)."""
        if len(found_objects) != len(meta_infos):
            raise AssertionError("one meta_info per image is required")
        rescaled = []
        for objects, meta in zip(found_objects, meta_infos):
            rescaled.append([obj.create_same_markup(utils.rescale_bbox(obj.bbox, xscale=meta.xscale, yscale=meta.yscale))
                             for obj in objects])
        return rescaled
