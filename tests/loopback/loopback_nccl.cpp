// TEST INFRASTRUCTURE: an in-process stand-in for the six RCCL entry points libubd_hip.so resolves with dlsym
// (ubdvss_amd/csrc/comm.hip), so that the N > 1 paths of the data-parallel train step -- the fused two-segment gradient
// all-reduce on the communication stream, the batch-global loss (histogram all-reduces level by level, tie-count all-gather),
// the parameter broadcast -- can EXECUTE on the one-GPU test box: the N ranks are N host threads of one process, each with
// its own ubd handle and stream on the same device.  Selected with UBD_RCCL_LIB=<this library> (comm.hip open_rccl).
//
// Semantics kept from NCCL: collectives are matched by call order per communicator; the result is the SUM over ranks in rank
// order (deterministic), visible to work enqueued on `stream` after the call.  Simplification: the call blocks the host
// thread until all ranks have arrived and synchronises `stream` first.  That is NOT a stricter test than NCCL's stream-ordered
// enqueue -- it is a weaker one for ordering: a missing cross-stream dependency (e.g. between the fused all-reduce on the
// communication stream, the stem backward on the caller's stream and ubd_comm_finish) is MASKED by the host-side
// synchronisation.  These tests therefore check the collective ARITHMETIC and call matching (who sums what, in which order,
// with which counts), not event ordering or overlap; the first real exercise of the ordering is bench.py's N > 1 leg on RCCL.
// Never linked into or loaded by the product unless UBD_RCCL_LIB names it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string.h>
#include <vector>

namespace {
struct group {
    int world = 0, arrived = 0, generation = 0, err = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::vector<char>> slot;     // per-rank contribution of the collective in flight
    std::vector<char> result;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct comm { group *g; int rank; };
std::mutex g_mu;
std::map<unsigned long long, group *> g_groups;
unsigned long long g_next_id = 1;

size_t dsize(ncclDataType_t t)
{
    switch (t) {
    case ncclFloat32: case ncclInt32: case ncclUint32: return 4;
    case ncclFloat64: case ncclInt64: case ncclUint64: return 8;
    default: return 0;
    }
}
template <typename T> void sum_into(std::vector<char> &res, const std::vector<std::vector<char>> &slot, size_t count)
{
    T *r = (T *)res.data();
    for (size_t i = 0; i < count; ++i) {
        T acc = ((const T *)slot[0].data())[i];
        for (size_t k = 1; k < slot.size(); ++k) acc += ((const T *)slot[k].data())[i];       // rank order: deterministic
        r[i] = acc;
    }
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof(*id));
    const unsigned long long v = g_next_id++;
    memcpy(id->internal, &v, sizeof(v));
    memcpy(id->internal + 8, "LOOPBACK", 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    unsigned long long v;
    memcpy(&v, id.internal, sizeof(v));
    if (memcmp(id.internal + 8, "LOOPBACK", 8) != 0 || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    group *g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_groups.find(v);
        if (it == g_groups.end()) {
            g = new group; g->world = nranks; g->slot.resize(nranks);
            g_groups[v] = g;
        } else g = it->second;
        if (g->world != nranks) return ncclInvalidArgument;
    }
    g->barrier();                                       // collective like the real one: returns once every rank has joined
    comm *c = new comm{g, rank};
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t cm) { delete (comm *)cm; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "loopback collective error"; }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t cm, hipStream_t st)
{
    comm *c = (comm *)cm; group *g = c->g;
    const size_t bytes = count * dsize(dt);
    if (op != ncclSum || bytes == 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    g->slot[c->rank].resize(bytes);
    if (hipMemcpy(g->slot[c->rank].data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    g->barrier();
    if (c->rank == 0) {
        g->err = 0;
        for (auto &s : g->slot) if (s.size() != bytes) g->err = 1;                      // mismatched collective across ranks
        g->result.assign(bytes, 0);
        if (!g->err) switch (dt) {
        case ncclFloat32: sum_into<float>(g->result, g->slot, count); break;
        case ncclFloat64: sum_into<double>(g->result, g->slot, count); break;
        case ncclInt32: sum_into<int>(g->result, g->slot, count); break;
        case ncclUint32: sum_into<unsigned>(g->result, g->slot, count); break;
        default: g->err = 1;
        }
    }
    g->barrier();
    if (g->err) return ncclInvalidArgument;             // every rank sees it: nobody is left waiting
    if (hipMemcpy(recv, g->result.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g->barrier();                                       // nobody overwrites `result` / the slots before everyone has read
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t cm, hipStream_t st)
{
    comm *c = (comm *)cm; group *g = c->g;
    const size_t bytes = sendcount * dsize(dt);
    if (bytes == 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    g->slot[c->rank].resize(bytes);
    if (hipMemcpy(g->slot[c->rank].data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    g->barrier();
    for (int k = 0; k < g->world; ++k) {
        if (g->slot[k].size() != bytes) continue;       // mismatched call on rank k: its part stays unwritten, the test sees it
        if (hipMemcpy((char *)recv + (size_t)k * bytes, g->slot[k].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    g->barrier();
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t cm, hipStream_t st)
{
    comm *c = (comm *)cm; group *g = c->g;
    const size_t bytes = count * dsize(dt);
    if (bytes == 0 || root < 0 || root >= g->world) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root) {
        g->result.resize(bytes);
        if (hipMemcpy(g->result.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    }
    g->barrier();
    if (hipMemcpy(recv, g->result.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g->barrier();
    return ncclSuccess;
}
}
