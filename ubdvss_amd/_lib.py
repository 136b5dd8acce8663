"""ctypes binding of libubd_hip.so (include/ubd.h).  Fails loudly when the library is absent."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UBD_LIB_PATH: another build of the same library (diagnostic builds of tools/: stamps, the racy-flatten proof); never a fallback
LIB_PATH = os.environ.get("UBD_LIB_PATH") or os.path.join(_HERE, "libubd_hip.so")

UBD_F32, UBD_BF16, UBD_F16 = 0, 1, 2
UBD_IN_F32, UBD_IN_U8 = 0, 1
UBD_IN_PREPACKED = 0x100
UBD_PRE_NONE, UBD_PRE_MOBILENET = 0, 1
UBD_COMM_FUSED = 1
UBD_COMM_GLOBAL_LOSS = 2
UBD_UNIQUE_ID_BYTES = 128
ABI_VERSION = 3


class UbdConfig(ctypes.Structure):
    _fields_ = [("c_in", ctypes.c_int32), ("n_classes", ctypes.c_int32),
                ("fml_compatible", ctypes.c_int32), ("dtype", ctypes.c_int32)]


# every symbol include/ubd.h declares: name -> (restype, argtypes)
_vp, _sz, _i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
_f = ctypes.c_float
SIGNATURES = {
    "ubd_abi_version": (_i, []),
    "ubd_build_id": (ctypes.c_char_p, []),
    "ubd_last_error": (ctypes.c_char_p, []),
    "ubd_host_memcpy_mt": (_i, [_vp, _vp, _sz, _i]),
    "ubd_create": (_i, [ctypes.POINTER(UbdConfig), ctypes.POINTER(_vp)]),
    "ubd_destroy": (None, [_vp]),
    "ubd_param_count": (_sz, [_vp]),
    "ubd_num_cus": (_i, [_vp]),
    "ubd_forward_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "ubd_train_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "ubd_postprocess_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "ubd_loss_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "ubd_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ubd_forward_postprocess": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz,
                                     _vp, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "ubd_pack_weights": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "ubd_dilated_layer": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ubd_postprocess": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "ubd_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ubd_train_step": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ubd_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _f, _f, _f, _f, _f, _vp]),
    "ubd_build_label_maps": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ubd_comm_unique_id": (_i, [_vp]),
    "ubd_comm_init": (_i, [_vp, _vp, _i, _i, _i]),
    "ubd_comm_destroy": (_i, [_vp]),
    "ubd_comm_world": (_i, [_vp]),
    "ubd_allreduce_grads": (_i, [_vp, _vp, _sz, _vp]),
    "ubd_broadcast_params": (_i, [_vp, _vp, _sz, _i, _vp]),
}

_lib = None


def load():
    """Returns the loaded library; raises (never falls back) when it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  ubdvss_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        lib.ubd_abi_version.restype = ctypes.c_int
        lib.ubd_abi_version.argtypes = []
        if lib.ubd_abi_version() != ABI_VERSION:              # checked BEFORE the symbol table: a stale build says so instead of an AttributeError
            raise RuntimeError(f"{LIB_PATH}: ABI {lib.ubd_abi_version()} != expected {ABI_VERSION}; rebuild (ubdvss_amd/csrc/build.sh)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError = stale build, surface it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {load().ubd_last_error().decode()}")


class StreamWorkspaces:
    """Device scratch per (device, stream) for the static entry points (losses.get_loss, SegmapManager.postprocess): calls on one
    stream are ordered and may share a scratch, two streams never do.  Bounded: the least recently used entry goes when more than
    ``max_entries`` streams have been seen (a process that makes many short-lived streams would otherwise keep one full-size
    scratch per stream handle for ever; a dropped tensor returns to torch's caching allocator, which reuses it in stream order)."""

    def __init__(self, max_entries=8):
        import collections
        self._d = collections.OrderedDict()
        self._max = int(max_entries)

    def get(self, device, raw_stream, nbytes):
        import torch
        key = (str(device), int(raw_stream))
        ws = self._d.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self._d[key] = ws
        self._d.move_to_end(key)
        while len(self._d) > self._max:
            self._d.popitem(last=False)
        return ws

    def __len__(self):
        return len(self._d)
