"""ModelRunner.predict -- host mirror of semantic_segmentation/model_runner.py:31-38, :105-148.

The reference runs ``model.predict`` on the device, then thresholds with numpy and calls
``SegmapManager.postprocess`` (OpenCV) image by image on the host.  Here the forward pass, the
threshold, the component labelling and the box fitting all run on the MI355X; only the final
(count, quads, classes) lists cross PCIe.
"""
import ctypes
import os
import numpy as np
import torch

from . import _lib, utils
from .data_markup import ObjectMarkup, ClassifiedObjectMarkup


class _DeviceEvent:
    """A HIP event created with hipEventDisableTiming | hipEventDisableSystemFence: recording it orders work between two
    streams of THIS device without the system-scope cache write-back a default event carries (the forward stream goes straight
    on with the next batch; nothing on the host reads what the event guards).  torch.cuda.Event cannot express the second
    flag, so the runtime is called directly; same three operations the pipeline needs (record / wait on a stream / query)."""
    _hip = None
    _FLAGS = 0x2 | 0x20000000            # hipEventDisableTiming | hipEventDisableSystemFence

    def __init__(self):
        cls = type(self)
        if cls._hip is None:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
            hip.hipEventQuery.argtypes = [ctypes.c_void_p]
            hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
            hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
            cls._hip = hip
        self._e = ctypes.c_void_p()
        rc = cls._hip.hipEventCreateWithFlags(ctypes.byref(self._e), cls._FLAGS)
        if rc != 0:
            raise RuntimeError(f"hipEventCreateWithFlags failed ({rc})")

    def record(self, stream):
        rc = self._hip.hipEventRecord(self._e, ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed ({rc})")

    def wait(self, stream):
        rc = self._hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), self._e, 0)
        if rc != 0:
            raise RuntimeError(f"hipStreamWaitEvent failed ({rc})")

    def query(self):
        return self._hip.hipEventQuery(self._e) == 0

    def synchronize(self):
        rc = self._hip.hipEventSynchronize(self._e)
        if rc != 0:
            raise RuntimeError(f"hipEventSynchronize failed ({rc})")

    def __del__(self):
        e, self._e = getattr(self, "_e", None), None
        if e and self._hip is not None:
            self._hip.hipEventDestroy(e)


class _TorchEvent:
    """torch.cuda.Event (default flags: its record carries a system-scope release) behind the same calls.  Used for the
    "results complete" event -- the HOST reads those results after waiting for it -- and, with UBD_PIPE_EVENTS=torch, for the
    device-to-device one too (A/B of the event flavour)."""

    def __init__(self):
        self._e = torch.cuda.Event()

    def record(self, stream):
        self._e.record(stream)

    def wait(self, stream):
        stream.wait_event(self._e)

    def query(self):
        return self._e.query()

    def synchronize(self):
        self._e.synchronize()


class ModelRunner:
    def __init__(self, net_config, pixel_threshold=0.5, max_objects_per_image=256, pipelined=False):
        """model_runner.py:31-38: pixel_probability > pixel_threshold is positive.
        pipelined=True: the postprocess of batch k runs on a second HIP stream and overlaps the forward pass
        of batch k+1 (double-buffered logits / results); results of a call are complete once
        ``self.last_event`` has been waited for (``synchronize()``)."""
        self._net_config = net_config
        eps = 1e-9
        self._logit_threshold = - np.log(1 / np.clip(pixel_threshold, eps, 1 - eps) - 1)
        self._cap = max_objects_per_image
        self._pipelined = pipelined
        self._stagger_us = int(os.environ.get("UBD_STAGGER_US", "0"))
        self._event_cls = _TorchEvent if os.environ.get("UBD_PIPE_EVENTS", "hip") == "torch" else _DeviceEvent
        self._side = None
        self._slots = {}
        self._step = 0
        self.last_event = None

    def synchronize(self):
        if self.last_event is not None:
            self.last_event.synchronize()

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
        # ---- two-stream pipeline: forward on the caller's stream, postprocess on the side stream
        n, hh, ww, _ = images.shape
        key = (self._step & 1, n, hh, ww, id(model))
        self._step += 1
        main = torch.cuda.current_stream(model.device)
        if self._side is None:
            # high priority: the short, latency-bound postprocess kernels take their few CUs at once instead of queueing
            # behind the next forward pass (+2 % end to end)
            self._side = torch.cuda.Stream(device=model.device, priority=-1)
        slot = self._slots.get(key)
        if slot is None:
            slot = {"logits": torch.empty((n, hh // 4, ww // 4, model.k_out), dtype=torch.float32, device=model.device),
                    "out": model.alloc_postprocess_outputs(n, hh // 4, ww // 4, self._cap), "done": None, "used": False,
                    "fwd": self._event_cls()}               # logits ready: consumed by kernels of this device only
            self._slots[key] = slot
        if slot["used"] and not slot["done"].query():
            slot["done"].wait(main)                     # the previous postprocess of this slot still reads its logits
                                                        # (normally long finished: no barrier packet in the forward stream)
        if self.last_event is not None and self._stagger_us > 0:
            # the postprocess of the previous batch was enqueued on the side stream a moment ago: give its whole-CU blocks a
            # head start over the 16 384 small blocks of the first stem kernel (ubd_stream_delay, include/ubd.h)
            _lib.check(_lib.load().ubd_stream_delay(ctypes.c_void_p(main.cuda_stream), self._stagger_us), "ubd_stream_delay")
        logits = model.predict_on_device(images, out=slot["logits"])
        slot["fwd"].record(main)                        # the slot's events are reused: its previous postprocess has been waited for above
        slot["fwd"].wait(self._side)
        if slot["done"] is None:
            slot["done"] = _TorchEvent()                # results ready: the host reads them after this one
        with torch.cuda.stream(self._side):
            bmap, quads, classes, counts = model.postprocess_on_device(logits, self._logit_threshold, scale, min_area,
                                                                       cap=self._cap, outputs=slot["out"])
            slot["done"].record(self._side)
        slot["used"] = True
        self.last_event = slot["done"]
        return logits, bmap, quads, classes, counts

    def predict(self, model, images, rescale=False, meta_infos=None):
        """Same contract as the reference (model_runner.py:105-138): returns
        (detection map (N,h,w,1) of {0,1}, classification_logits (N,h,w,n_cls), found_objects)."""
        assert not rescale or (meta_infos is not None and len(images) == len(meta_infos))
        x = np.asarray(images)
        if x.dtype != np.uint8:
            x = x.astype(np.float32)
        # float images arrive already preprocessed, as in the reference; uint8 images are raw pixels and take the
        # NetConfig preprocessing fused into the first layer (one rule for every entry point)
        xt = torch.from_numpy(np.ascontiguousarray(x)).to(model.device)
        logits, bmap, quads, classes, counts = self.predict_on_device(model, xt)
        self.synchronize()
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
