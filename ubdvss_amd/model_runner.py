"""ModelRunner.predict -- host mirror of semantic_segmentation/model_runner.py:31-38, :105-148.

The reference runs ``model.predict`` on the device, then thresholds with numpy and calls
``SegmapManager.postprocess`` (OpenCV) image by image on the host.  Here the forward pass, the
threshold, the component labelling and the box fitting all run on the MI355X; only the final
(count, quads, classes) lists cross PCIe.
"""
import os

import numpy as np
import torch

from . import _lib, utils
from .data_markup import ObjectMarkup, ClassifiedObjectMarkup


class ModelRunner:
    def __init__(self, net_config, pixel_threshold=0.5, max_objects_per_image=256, pipelined=False, slots=2):
        """model_runner.py:31-38: pixel_probability > pixel_threshold is positive.
        pipelined=True: the postprocess of batch k is enqueued together with the forward pass of batch k+1 -- inside the
        first blocks of that pass's stem kernel (ubd_forward_postprocess), on the same stream, without events.  The result
        tensors a call returns are therefore COMPLETE only after the next call (or ``flush()``) has been enqueued and the
        stream has been waited for; ``synchronize()`` does both.  Logits / results are double-buffered: they stay valid until
        the second call after the one that returned them (``slots`` = 3: until the third -- ``predict_stream`` copies a batch's
        results to the host while the next two batches are in flight)."""
        self._net_config = net_config
        eps = 1e-9
        self._logit_threshold = - np.log(1 / np.clip(pixel_threshold, eps, 1 - eps) - 1)
        self._cap = max_objects_per_image
        self._pipelined = pipelined
        self._nslots = max(2, int(slots))
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
        key = (self._step % self._nslots, n, hh, ww, id(model))
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
        return self._assemble(counts.cpu().numpy(), quads.cpu().numpy(), classes.cpu().numpy() if classes is not None else None,
                              logits.cpu().numpy(), bmap.cpu().numpy(), rescale, meta_infos)

    def _assemble(self, counts_h, quads_h, classes_h, logits_h, bmap_h, rescale, meta_infos, copy=False):
        """Host arrays -> the reference's return triple (model_runner.py:121-138).  copy: the arrays are reused staging buffers."""
        if (counts_h > self._cap).any():
            raise RuntimeError(f"more than max_objects_per_image={self._cap} objects in an image "
                               f"(max found {int(counts_h.max())}); raise the capacity")
        detection_logits = bmap_h.astype(np.int64)[..., None]
        classification_logits = np.array(logits_h[..., 1:]) if copy else logits_h[..., 1:]
        with_cls = self._net_config.is_classification_supported()
        found_objects = []
        for i in range(len(counts_h)):
            n = int(counts_h[i])
            boxes = quads_h[i, :n].astype(int)              # one conversion per image (a fresh array: the rows are views of it)
            if with_cls:
                found_objects.append([ClassifiedObjectMarkup(boxes[j], classes_h[i, j]) for j in range(n)])
            else:
                found_objects.append([ObjectMarkup(boxes[j]) for j in range(n)])
        if rescale:
            found_objects = self.rescale(found_objects, meta_infos)
        return detection_logits, classification_logits, found_objects

    def predict_stream(self, model, batches, rescale=False, meta_infos=None, copy_threads=8):
        """The loop of the reference's ``ModelRunner.run`` (model_runner.py:60-67: ``predict`` batch after batch) as ONE pipeline:
        generator over ``predict``'s return triple, one per batch of ``batches`` (an iterable of numpy (N,H,W,C) arrays, uint8 or
        float32), in order.  Batch k + 1 is staged and transferred while batch k computes:
          staging thread: the batch is copied into one of three PINNED staging buffers by ``copy_threads`` native threads (one foreign call,
            ``ubd_host_memcpy_mt``) -- the transfer of pageable memory would otherwise be staged by the runtime, synchronously;
          copy-in stream: pinned -> device (three device input buffers), enqueued as soon as the batch is staged and BEFORE the
            consumer blocks on an older batch's results; an event hands the buffer to the compute stream;
          compute stream: the pipelined runner -- forward pass of batch k with the postprocess of batch k - 1 inside its stem kernel;
          copy-out stream: (logits, map, quads, classes, counts) of batch k - 1 -> pinned host memory behind an event;
          consumer (this generator): the object lists of batch k - 2 are built while all of that is in flight.
        ``meta_infos``: one list per batch when ``rescale``.  Results are identical to calling ``predict`` per batch
        (tests/test_gpu_end_to_end.py::test_predict_stream_equals_predict)."""
        import queue
        import threading
        import time
        dev = model.device
        runner = ModelRunner(self._net_config, max_objects_per_image=self._cap, pipelined=True, slots=3)
        runner._logit_threshold = self._logit_threshold
        s_main = torch.cuda.current_stream(dev)
        s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        lib = _lib.load()
        try:                                            # never more staging threads than half of the CPUs this process may run on
            copy_threads = max(1, min(int(copy_threads), len(os.sched_getaffinity(0)) // 2))
        except (AttributeError, OSError):
            copy_threads = max(1, int(copy_threads))
        NPIN, NDEV, NRES = 3, 3, 3
        pinned, dev_in, host_res = [None] * NPIN, [None] * NDEV, [None] * NRES
        fwd_done, d2h_done = [None] * NDEV, [None] * NRES
        inflight = []                                   # [batch index, device results, d2h event or None, meta]
        if rescale and meta_infos is None:
            raise AssertionError("rescale=True needs meta_infos (one list of meta_info per batch)")     # predict()'s rule (model_runner.py:116-117)
        metas = iter(meta_infos) if meta_infos is not None else None
        _MISSING = object()
        free_slots, staged = queue.Queue(), queue.Queue(maxsize=NPIN - 1)
        for slot in range(NPIN):
            free_slots.put((slot, None))
        stop = threading.Event()
        stats = {"wait_staged_s": 0.0, "enqueue_s": 0.0, "deliver_s": 0.0, "deliver_wait_s": 0.0, "stager_copy_s": 0.0, "batches": 0}
        self.last_stream_stats = stats                  # where the consumer thread's time went (tools/bench_host_stream.py prints it)

        def stager():
            """host side of the copy-in, its own thread: batch -> pinned staging buffer, ahead of the stream work"""
            try:
                torch.cuda.set_device(dev)              # a new thread starts on cuda:0: pinned allocations / event waits belong to the model's device
                for k, x in enumerate(batches):
                    if stop.is_set():
                        return
                    x = np.asarray(x)
                    if x.dtype != np.uint8:
                        x = x.astype(np.float32, copy=False)
                    x = np.ascontiguousarray(x)
                    tdt = torch.uint8 if x.dtype == np.uint8 else torch.float32
                    slot, ev = free_slots.get()
                    if ev is not None:
                        ev.synchronize()                # the transfer that last read this staging buffer is over
                    if pinned[slot] is None or tuple(pinned[slot].shape) != x.shape or pinned[slot].dtype != tdt:
                        pinned[slot] = torch.empty(x.shape, dtype=tdt, pin_memory=True)
                    dst = pinned[slot].numpy()
                    tc = time.perf_counter()
                    # one foreign call (ctypes drops the GIL), native threads inside: as `copy_threads` python pool tasks the copy took the
                    # interpreter away from the consumer thread at every hand-over (enqueue 0.15 -> 0.35 -> 0.8 ms at 2 / 4 / 8 threads)
                    lib.ubd_host_memcpy_mt(dst.ctypes.data, x.ctypes.data, x.nbytes, int(copy_threads))
                    stats["stager_copy_s"] += time.perf_counter() - tc       # the staging thread's own time (not the consumer's)
                    meta = None
                    if metas is not None:
                        meta = next(metas, _MISSING)
                        if meta is _MISSING:
                            raise ValueError(f"meta_infos ended before the batches did (batch {k} has none)")
                    staged.put((k, slot, meta))
                staged.put(None)
            except Exception as e:                      # noqa: BLE001 -- re-raised by the consumer
                staged.put(e)

        def copy_in(item):
            """pinned -> device on the copy-in stream; returns (k, device buffer index, event, meta)"""
            k, slot, meta = item
            d, src = k % NDEV, pinned[slot]
            if dev_in[d] is None or dev_in[d].shape != src.shape or dev_in[d].dtype != src.dtype:
                dev_in[d] = torch.empty(src.shape, dtype=src.dtype, device=dev)
            with torch.cuda.stream(s_in):
                if fwd_done[d] is not None:
                    s_in.wait_event(fwd_done[d])        # the forward pass that last read this device buffer is over
                dev_in[d].copy_(src, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(s_in)
            free_slots.put((slot, ev))                  # the stager waits for the event before it writes the buffer again
            return k, d, ev, meta

        def fetch(entry):
            """device results of batch entry[0] -> its pinned host set on the copy-out stream, behind everything enqueued on the compute stream so far"""
            r, res = entry[0] % NRES, entry[1]
            if host_res[r] is None or any((h is None) != (t is None) or (t is not None and (h.shape != t.shape or h.dtype != t.dtype)) for h, t in zip(host_res[r], res)):
                host_res[r] = [None if t is None else torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in res]
            ev_c = torch.cuda.Event(); ev_c.record(s_main)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_c)
                for h, t in zip(host_res[r], res):
                    if t is not None:
                        h.copy_(t, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(s_out)
            d2h_done[r] = ev
            entry[2] = ev

        def deliver(entry):
            tw = time.perf_counter()
            entry[2].synchronize()
            stats["deliver_wait_s"] += time.perf_counter() - tw     # of deliver_s: blocked on the device (results not there yet)
            logits_h, bmap_h, quads_h, classes_h, counts_h = [None if h is None else h.numpy() for h in host_res[entry[0] % NRES]]
            return self._assemble(counts_h, quads_h, classes_h, logits_h, bmap_h, rescale, entry[3], copy=True)

        th = threading.Thread(target=stager, daemon=True)
        th.start()
        try:
            ahead = None                                # batch k + 1, already on its way to the device
            done = False
            while True:
                t0 = time.perf_counter()
                if ahead is None:
                    if done:
                        break
                    item = staged.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    ahead = copy_in(item)
                k, d, ev, meta = ahead
                ahead = None
                t1 = time.perf_counter()
                stats["wait_staged_s"] += t1 - t0
                s_main.wait_event(ev)
                if d2h_done[k % NRES] is not None:
                    s_main.wait_event(d2h_done[k % NRES])           # the runner's result slot k % 3 is about to be overwritten: batch k - 3 has left it (long ago)
                res = runner.predict_on_device(model, dev_in[d])   # forward of batch k + postprocess of batch k - 1, one stream
                fe = torch.cuda.Event(); fe.record(s_main)
                fwd_done[d] = fe
                if inflight:                                        # batch k - 1 is complete behind this call: copy its results out
                    fetch(inflight[-1])
                inflight.append([k, res, None, meta])
                if not done:                                        # batch k + 1: start its transfer NOW, before blocking on older results
                    try:
                        item = staged.get_nowait()
                        if item is None:
                            done = True
                        elif isinstance(item, BaseException):
                            raise item
                        else:
                            ahead = copy_in(item)
                    except queue.Empty:
                        pass
                t2 = time.perf_counter()
                stats["enqueue_s"] += t2 - t1
                stats["batches"] += 1
                while len(inflight) > 2:                            # build batch k - 2's lists while k - 1 copies out and k computes
                    out = deliver(inflight.pop(0))
                    stats["deliver_s"] += time.perf_counter() - t2
                    yield out
            if inflight:
                runner.flush()                                      # the last batch's postprocess, a launch of its own
                fetch(inflight[-1])
            while inflight:
                yield deliver(inflight.pop(0))
        finally:
            stop.set()
            # An early exit (the consumer stopped, the stager failed, _assemble raised) can leave a prefetched host -> device copy or a
            # device -> host copy in flight on the side streams.  The buffers they touch were allocated on the compute stream: once this
            # frame drops them the caching allocator may hand the blocks to the next allocation on that stream, and the late copy would
            # write into an unrelated tensor.  Drain both side streams before anything goes out of scope.
            s_in.synchronize()
            s_out.synchronize()
            for _ in range(40):                                     # a consumer that stopped early: unblock the stager (bounded: the thread is a
                if not th.is_alive():                               # daemon, and one that is blocked inside next(batches) cannot see `stop`)
                    break
                try:
                    staged.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)

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
