// Data-parallel exchange step of the train step inside the C ABI: RCCL (over xGMI) on one flat fp32 gradient vector.
//
// The reference is single-device (no collective call sites); data parallelism is what the MI355X build adds
// (BASELINE.json configs[3]): one process per GPU, per-replica loss, ONE sum all-reduce of the 132 KB flat
// gradient vector per step, 1/world folded into the Adam kernel.  A non-torch caller gets that here: the handle owns an RCCL
// communicator (rank 0 creates the unique id with ubd_comm_unique_id and hands the 128 bytes to the other ranks by any
// means -- file, socket, the launcher's store), a communication stream and two events.
// Overlap: with UBD_COMM_FUSED, ubd_train_step all-reduces the gradients itself: the dilated + head segment (31 273 of the
// 33 028 floats, complete once the dilated layers' backward is done) goes out on the communication stream UNDER the
// backward pass of the three stem layers (0.5 ms of the 1.6 ms bf16 step); the 1 755 stem floats follow on the caller's
// stream, which then waits for the first part.
// librccl is resolved with dlopen at ubd_comm_init time (the copy already mapped by the process -- torch's -- wins; UBD_RCCL_LIB
// names another one), so the library has no link-time dependency on it and loads on hosts without RCCL / without a GPU.
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>
#include "common.h"

struct ubd_comm {
    void *lib;
    ncclComm_t comm;
    int rank, world, flags;
    hipStream_t stream;          // communication stream (overlap)
    hipEvent_t ready, done;      // segment A ready on the compute stream / reduced on the communication stream
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    const char *(*GetErrorString)(ncclResult_t);
};

static void *open_rccl()
{
    // UBD_RCCL_LIB: the collective library to use instead of the process's librccl -- a site's own RCCL build, or the in-process
    // loopback stand-in the tests use to run N ranks as N threads on one GPU (tests/loopback/)
    if (const char *over = getenv("UBD_RCCL_LIB")) {
        if (over[0]) return dlopen(over, RTLD_NOW | RTLD_LOCAL);
    }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        void *l = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);          // a copy the process has already mapped (torch's) first
        if (l) return l;
    }
    for (const char *nm : names) {
        void *l = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (l) return l;
    }
    return nullptr;
}

#define UBD_CHECK_NCCL(c, expr)                                                                       \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            ubd_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, (c)->GetErrorString ? (c)->GetErrorString(_r) : "?"); \
            return 3;                                                                                \
        }                                                                                            \
    } while (0)

extern "C" int ubd_comm_unique_id(void *id_out)
{
    UBD_REQUIRE(id_out, "ubd_comm_unique_id: null argument");
    void *lib = open_rccl();
    UBD_REQUIRE(lib, "ubd_comm_unique_id: librccl not found (%s)", dlerror());
    auto get = (ncclResult_t(*)(ncclUniqueId *))dlsym(lib, "ncclGetUniqueId");
    UBD_REQUIRE(get, "ubd_comm_unique_id: ncclGetUniqueId not found in librccl");
    ncclUniqueId id;
    ncclResult_t r = get(&id);
    UBD_REQUIRE(r == ncclSuccess, "ubd_comm_unique_id: ncclGetUniqueId failed (%d)", (int)r);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int ubd_comm_init(ubd_handle *h, const void *unique_id, int rank, int world, int flags)
{
    UBD_REQUIRE(h && unique_id, "ubd_comm_init: null argument");
    UBD_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ubd_comm_init: rank %d / world %d out of range", rank, world);
    UBD_REQUIRE(!h->comm, "ubd_comm_init: the handle already has a communicator");
    ubd_comm *c = (ubd_comm *)calloc(1, sizeof(ubd_comm));
    UBD_REQUIRE(c, "ubd_comm_init: out of host memory");
    c->lib = open_rccl();
    if (!c->lib) { ubd_set_error("ubd_comm_init: librccl not found (%s)", dlerror()); free(c); return 2; }
    auto init = (ncclResult_t(*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(c->lib, "ncclCommInitRank");
    c->AllReduce = (decltype(c->AllReduce))dlsym(c->lib, "ncclAllReduce");
    c->Broadcast = (decltype(c->Broadcast))dlsym(c->lib, "ncclBroadcast");
    c->AllGather = (decltype(c->AllGather))dlsym(c->lib, "ncclAllGather");
    c->CommDestroy = (decltype(c->CommDestroy))dlsym(c->lib, "ncclCommDestroy");
    c->GetErrorString = (decltype(c->GetErrorString))dlsym(c->lib, "ncclGetErrorString");
    if (!init || !c->AllReduce || !c->Broadcast || !c->AllGather || !c->CommDestroy) { ubd_set_error("ubd_comm_init: librccl lacks a required symbol"); free(c); return 2; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = init(&c->comm, world, id, rank);
    if (r != ncclSuccess) { ubd_set_error("ubd_comm_init: ncclCommInitRank failed: %s", c->GetErrorString ? c->GetErrorString(r) : "?"); free(c); return 3; }
    c->rank = rank; c->world = world; c->flags = flags;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) { ubd_set_error("ubd_comm_init: stream / event creation failed: %s", hipGetErrorString(e)); c->CommDestroy(c->comm); free(c); return 1; }
    h->comm = c;
    return 0;
}

extern "C" int ubd_comm_destroy(ubd_handle *h)
{
    UBD_REQUIRE(h, "ubd_comm_destroy: null argument");
    ubd_comm *c = h->comm;
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    c->CommDestroy(c->comm);
    (void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done); (void)hipStreamDestroy(c->stream);
    free(c);
    h->comm = nullptr;
    return 0;
}

extern "C" int ubd_comm_world(const ubd_handle *h) { return (h && h->comm) ? h->comm->world : 1; }

extern "C" int ubd_allreduce_grads(ubd_handle *h, float *grads, size_t count, void *stream)
{
    UBD_REQUIRE(h && grads, "ubd_allreduce_grads: null argument");
    UBD_REQUIRE(h->comm, "ubd_allreduce_grads: the handle has no communicator (ubd_comm_init first)");
    ubd_comm *c = h->comm;
    UBD_CHECK_NCCL(c, c->AllReduce(grads, grads, count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
    return 0;
}

extern "C" int ubd_broadcast_params(ubd_handle *h, float *params, size_t count, int root, void *stream)
{
    UBD_REQUIRE(h && params, "ubd_broadcast_params: null argument");
    UBD_REQUIRE(h->comm, "ubd_broadcast_params: the handle has no communicator (ubd_comm_init first)");
    ubd_comm *c = h->comm;
    UBD_REQUIRE(root >= 0 && root < c->world, "ubd_broadcast_params: root %d out of range", root);
    UBD_CHECK_NCCL(c, c->Broadcast(params, params, count, ncclFloat32, root, c->comm, (hipStream_t)stream));
    return 0;
}

// ---- hooks of the loss (loss.hip) with UBD_COMM_GLOBAL_LOSS ---------------------------------------------------------------
bool ubd_comm_global_loss(const ubd_handle *h) { return h->comm && (h->comm->flags & UBD_COMM_GLOBAL_LOSS); }
int ubd_comm_rank(const ubd_handle *h) { return h->comm ? h->comm->rank : 0; }

int ubd_comm_allreduce_raw(ubd_handle *h, void *buf, size_t count, int kind, hipStream_t st)
{
    ubd_comm *c = h->comm;
    const ncclDataType_t dt = kind == UBD_RED_F64 ? ncclFloat64 : (kind == UBD_RED_I32 ? ncclInt32 : ncclUint32);
    UBD_CHECK_NCCL(c, c->AllReduce(buf, buf, count, dt, ncclSum, c->comm, st));
    return 0;
}

int ubd_comm_allgather_u32(ubd_handle *h, const unsigned *send_one, unsigned *recv_world, hipStream_t st)
{
    ubd_comm *c = h->comm;
    UBD_CHECK_NCCL(c, c->AllGather(send_one, recv_world, 1, ncclUint32, c->comm, st));
    return 0;
}

// ---- hooks of ubd_train_step (backward.hip) with UBD_COMM_FUSED ----------------------------------------------------------
bool ubd_comm_fused(const ubd_handle *h) { return h->comm && (h->comm->flags & UBD_COMM_FUSED); }

// the dilated + head gradients [off_dil_k[0], n_params) are final: reduce them on the communication stream
int ubd_comm_begin_tail(ubd_handle *h, float *grads, hipStream_t st)
{
    ubd_comm *c = h->comm;
    const size_t first = h->off_dil_k[0];
    UBD_CHECK_HIP(hipEventRecord(c->ready, st));
#if !(defined(UBD_SABOTAGE_COMM) && UBD_SABOTAGE_COMM == 1)   // diagnostic build of tools/prove_comm_ordering.sh: this wait dropped, the ordering test must go red
    UBD_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
#endif
    UBD_CHECK_NCCL(c, c->AllReduce(grads + first, grads + first, h->n_params - first, ncclFloat32, ncclSum, c->comm, c->stream));
    UBD_CHECK_HIP(hipEventRecord(c->done, c->stream));
    return 0;
}

// the stem gradients [0, off_dil_k[0]) are final: reduce them on the caller's stream, then join the first part
int ubd_comm_finish(ubd_handle *h, float *grads, hipStream_t st)
{
    ubd_comm *c = h->comm;
    UBD_CHECK_NCCL(c, c->AllReduce(grads, grads, h->off_dil_k[0], ncclFloat32, ncclSum, c->comm, st));
#if !(defined(UBD_SABOTAGE_COMM) && UBD_SABOTAGE_COMM == 2)
    UBD_CHECK_HIP(hipStreamWaitEvent(st, c->done, 0));
#endif
    return 0;
}
