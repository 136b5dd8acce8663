"""Data-parallel plumbing (the reference is single-device).

One process per GPU.  Inference shards images by rank with no collective (replicas).  Training has
ONE exchange step per iteration: a sum all-reduce of the flat fp32 gradient vector (33 028 floats for
the RGB detection model) over RCCL/xGMI (``backend="nccl"`` on ROCm) -- or gloo on CPU in the tests;
the 1/world scaling is folded into the Adam kernel.  The loss is evaluated per replica.
"""
import os

import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
    """Initialises torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device(f"cuda:{local}")
        dist.init_process_group(backend, **kwargs)
    return rk, world, local


def shard_range(n_items, rk, world):
    """Contiguous [lo, hi) slice of n_items owned by rank rk (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)


def allreduce_gradients(flat_grads, group=None):
    """In-place SUM all-reduce of the flat gradient vector; returns the scale (1/world) the optimiser
    must apply to obtain the data-parallel mean."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w


def broadcast_parameters(flat_params, src=0, group=None):
    if world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def attach_native_comm(model, fused=True, group=None, global_loss=False):
    """Gives ``model``'s C-ABI handle its own RCCL communicator (include/ubd.h, ubd_comm_*): rank 0 creates the unique id,
    torch.distributed (already initialised, any backend) only carries those 128 bytes to the other ranks.  With ``fused``
    the train step all-reduces the gradients itself, overlapped with the stem layers' backward pass, and
    ``Trainer.apply_gradients`` skips the torch collective.  ``global_loss``: the loss reductions (n_pos, n_neg, means, the top-k
    of the flattened batch, losses.py:86-126) run over the images of all ranks -- the reference's loss at the global batch
    (option 2) -- and the gradients are summed, not averaged.  Returns the world size.  Also valid for a single
    process."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    w, r = world_size(group), rank(group)
    buf = (ctypes.c_char * _lib.UBD_UNIQUE_ID_BYTES)()
    err = None
    if r == 0:
        try:
            _lib.check(lib.ubd_comm_unique_id(buf), "ubd_comm_unique_id")
        except Exception as e:                          # noqa: BLE001 -- re-raised below, on every rank
            err = e
    if w > 1:
        # the id plus one status byte: the other ranks are already waiting in this broadcast, so rank 0 takes part in it even
        # when it could not create the id, and then every rank raises
        t = torch.frombuffer(bytearray(bytes(buf) + bytes([0 if err is None else 1])), dtype=torch.uint8).clone()
        if dist.get_backend(group) == "nccl":
            t = t.to(model.device)
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        ctypes.memmove(buf, raw[:_lib.UBD_UNIQUE_ID_BYTES], _lib.UBD_UNIQUE_ID_BYTES)
        if raw[_lib.UBD_UNIQUE_ID_BYTES] != 0 and err is None:
            err = RuntimeError("attach_native_comm: rank 0 could not create the RCCL unique id")
    if err is not None:
        raise err
    with torch.cuda.device(model.device):
        flags = (_lib.UBD_COMM_FUSED if fused else 0) | (_lib.UBD_COMM_GLOBAL_LOSS if global_loss else 0)
        _lib.check(lib.ubd_comm_init(model._h, buf, r, w, flags), "ubd_comm_init")
    model._native_comm = "fused" if fused else "explicit"
    model._global_loss = bool(global_loss)
    return w


def detach_native_comm(model):
    """Undoes attach_native_comm on this rank: destroys the handle's communicator (ubd_comm_destroy; harmless when there is
    none) so that ubd_train_step issues no collective of its own, and clears the Python-side markers.  When one rank failed to
    create its communicator EVERY rank must call this before falling back to torch.distributed's all-reduce -- otherwise the
    ranks that succeeded would block in RCCL calls the others never join (or sum the gradients twice)."""
    from . import _lib
    lib = _lib.load()
    with torch.cuda.device(model.device):
        _lib.check(lib.ubd_comm_destroy(model._h), "ubd_comm_destroy")
    model._native_comm = None
    model._global_loss = False


def max_over_ranks(value, device=None, group=None):
    """max of a python float over all ranks (bench timing contract)."""
    if world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
