"""ubdvss_amd -- MI355X-native (gfx950) hot path of asmekal/ubdvss.

Host-side mirror of the reference's Python seams (NetConfig / NetManager / Model.predict,
ModelRunner.predict, SegmapManager.postprocess, losses.get_loss, Adam train step) over the
C ABI of ``libubd_hip.so`` (include/ubd.h).  PyTorch is used only for device memory, streams
and ``torch.distributed``.  There is no CPU fallback: importing the kernels without the built
library, or creating a model without an MI355X, raises.
"""
from .data_markup import ObjectMarkup, ClassifiedObjectMarkup  # noqa: F401
from .net import NetConfig, NetManager, PreprocessingType, Model  # noqa: F401
from .model_runner import ModelRunner  # noqa: F401
from .segmap_manager import SegmapManager  # noqa: F401
from . import losses  # noqa: F401
from .trainer import Trainer, Adam  # noqa: F401
