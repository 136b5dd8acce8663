"""Train step -- host mirror of ``model.compile(Adam(lr), loss)`` + the per-step body of
``fit_generator`` (train.py:110-112, :176-188): forward, loss, backward, (data-parallel gradient
all-reduce), Adam.

Multi-GPU: one process per GPU; when ``torch.distributed`` is initialised the flat fp32 gradient
vector (33 028 floats for the RGB model) is summed with ONE all-reduce over RCCL/xGMI and divided
by the world size inside the fused Adam kernel.  The loss (incl. its batch-global top-k) is
evaluated per replica (option 1).
"""
import ctypes

import numpy as np
import torch

from . import _lib, distributed
from .net import PreprocessingType


class Adam:
    """keras.optimizers.Adam defaults (train.py:110): beta1 .9, beta2 .999, epsilon 1e-7."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon


class Trainer:
    def __init__(self, model, optimizer=None, process_group=None):
        self.model = model
        self.opt = optimizer or Adam()
        self._lib = _lib.load()
        n = model.count_params()
        dev = model.device
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(16, dtype=torch.float32, device=dev)
        self.iterations = 0
        self._ws = None
        self._pg = process_group

    def broadcast_weights(self, src=0):
        if getattr(self.model, "_native_comm", None):           # the handle's own RCCL communicator (ubd_broadcast_params)
            with torch.cuda.device(self.model.device):
                _lib.check(self._lib.ubd_broadcast_params(self.model._h, self.model.params.data_ptr(), self.model.params.numel(),
                                                          src, self.model._stream()), "ubd_broadcast_params")
        else:
            if self._pg is not False:
                distributed.broadcast_parameters(self.model.params, src=src, group=self._pg)
        # c10d writes the tensor without bumping its version counter: invalidate the packed weight fragments
        self.model.invalidate_packed_weights()

    def backward_on_device(self, images, targets):
        """images: device tensor (N,H,W,C) float32 or uint8; targets: device int32 (N,H/4,W/4[,1]).
        Leaves gradients in self.grads and [total, det, cls, k] in self.loss."""
        mdl = self.model
        images = images.contiguous()
        n, hh, ww, _ = images.shape
        targets = targets.reshape(n, hh // 4, ww // 4).to(torch.int32).contiguous()
        if images.dtype == torch.uint8:
            in_dtype = _lib.UBD_IN_U8
            pre = _lib.UBD_PRE_MOBILENET if mdl.net_config.get_preprocessing_type() == PreprocessingType.MOBILENET_LIKE else _lib.UBD_PRE_NONE
        else:
            in_dtype, pre = _lib.UBD_IN_F32, _lib.UBD_PRE_NONE
        nbytes = self._lib.ubd_train_workspace_bytes(mdl._h, n, hh, ww)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=mdl.device)
        with torch.cuda.device(mdl.device):
            _lib.check(self._lib.ubd_train_step(mdl._h, mdl.params.data_ptr(), images.data_ptr(), in_dtype, pre,
                                                targets.data_ptr(), n, hh, ww, self.grads.data_ptr(),
                                                self.loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                mdl._stream()), "ubd_train_step")

    def apply_gradients(self):
        native = getattr(self.model, "_native_comm", None)
        if native:                                              # C-ABI collective: fused into ubd_train_step, or explicit here
            world = self._lib.ubd_comm_world(self.model._h)
            if native == "explicit":
                with torch.cuda.device(self.model.device):
                    _lib.check(self._lib.ubd_allreduce_grads(self.model._h, self.grads.data_ptr(), self.grads.numel(),
                                                             self.model._stream()), "ubd_allreduce_grads")
            # per-replica losses are averaged; the batch-global loss (UBD_COMM_GLOBAL_LOSS) is ONE objective whose parameter
            # gradient is the sum of the ranks' contributions
            grad_scale = 1.0 if getattr(self.model, "_global_loss", False) else 1.0 / world
        elif self._pg is False:                                 # process_group=False: no collective at all, whatever torch.distributed holds
            grad_scale = 1.0                                    # (the timing reference of tools/dist_rccl_check.py: one GPU on its shard alone)
        else:
            grad_scale = distributed.allreduce_gradients(self.grads, group=self._pg)
        self.iterations += 1
        o = self.opt
        with torch.cuda.device(self.model.device):
            _lib.check(self._lib.ubd_adam_step(self.model.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                               self.v.data_ptr(), self.grads.numel(), self.iterations, o.lr, o.beta_1,
                                               o.beta_2, o.epsilon, grad_scale, self.model._stream()), "ubd_adam_step")
        self.model.invalidate_packed_weights()    # parameters changed behind torch's version counter

    def train_step_on_device(self, images, targets):
        self.backward_on_device(images, targets)
        self.apply_gradients()
        return self.loss

    def train_on_batch(self, images, targets):
        """Keras ``train_on_batch``: numpy batch in, scalar loss out.  Float images are fed as they are (already
        preprocessed, data_generators.py:113,148); uint8 images are raw pixels and get ``NetConfig``'s preprocessing
        fused into the first layer -- the same rule as every other entry point (Model.predict, ModelRunner.predict,
        *_on_device)."""
        dev = self.model.device
        x = np.asarray(images)
        if x.dtype != np.uint8:
            x = x.astype(np.float32)
        xt = torch.from_numpy(np.ascontiguousarray(x)).to(dev)     # uint8 stays uint8: NetConfig preprocessing fused on device
        yt = torch.from_numpy(np.ascontiguousarray(np.asarray(targets)).astype(np.int32)).to(dev)
        return float(self.train_step_on_device(xt, yt)[0].item())
