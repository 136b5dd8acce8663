// TEST INFRASTRUCTURE: an in-process, STREAM-ORDERED stand-in for the six RCCL entry points libubd_hip.so resolves with dlsym
// (ubdvss_amd/csrc/comm.hip), so that the N > 1 paths of the data-parallel train step -- the fused two-segment gradient
// all-reduce on the communication stream, the batch-global loss (histogram all-reduces level by level, tie-count all-gather),
// the parameter broadcast -- can EXECUTE on the one-GPU test box: the N ranks are N host threads of one process, each with
// its own ubd handle and streams on the same device.  Selected with UBD_RCCL_LIB=<this library> (comm.hip open_rccl).
//
// Semantics kept from NCCL -- including the visibility contract, so that a missing cross-stream dependency in the caller is NOT
// masked (round 4's version synchronised the stream on the host and said so):
//   * collectives are matched by call order per communicator; every rank must make the call (the host threads meet, which
//     delays the ENQUEUE only -- no stream is ever synchronised and no device buffer is read or written by the host);
//   * a rank's contribution is read by a device copy enqueued on the stream it passed: it sees what that stream's earlier work
//     wrote and NOTHING that is merely in flight on another stream;
//   * the result is the SUM over ranks in rank order (deterministic), produced by a device kernel on the group's own stream behind
//     the events of all ranks' contributions, and copied out on each rank's stream behind that kernel's event: it is visible to
//     work enqueued on `stream` after the call and to no other stream without an event.
// Buffers: a ring of GENERATIONS staging sets (per-rank slots + result) on the device; a set is reused only behind the events of
// its previous readers.  tests/test_gpu_comm_loopback.py proves the power of this with a sabotaged product build (one event wait
// dropped in comm.hip): the ordering test goes red (tools/prove_comm_ordering.sh, profiles/r05_comm_ordering_power.log).
// Never linked into or loaded by the product unless UBD_RCCL_LIB names it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
constexpr int GENERATIONS = 8;
constexpr int MAX_WORLD = 16;

struct group {
    int world = 0, arrived = 0, generation = 0, err = 0;
    std::mutex mu;
    std::condition_variable cv;
    long seq = 0;                                         // collectives completed (host side): picks the staging set
    size_t cap = 0;                                       // bytes per slot
    char *slots[GENERATIONS] = {};                        // [world][cap] contributions
    char *result[GENERATIONS] = {};
    hipStream_t gs[GENERATIONS] = {};                     // the group's streams, one per staging set: consecutive collectives are NOT ordered with
                                                          // each other by the stand-in (only by the callers' streams and events)
    hipEvent_t ev_in[GENERATIONS][MAX_WORLD] = {}, ev_out[GENERATIONS] = {}, ev_done[GENERATIONS][MAX_WORLD] = {};
    size_t posted[MAX_WORLD] = {};                        // bytes each rank brought to the collective in flight (mismatch check)
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

// LOOPBACK_DELAY_US (for all-reduces of at least LOOPBACK_DELAY_MIN_COUNT elements): the sum starts that many microseconds late (the
// wire's latency): a caller that consumes the result without waiting for the collective's stream then reads the OLD bytes for
// certain, not by luck of timing
__global__ void delay_kernel(long long ticks_100mhz)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
}

template <typename T> __global__ void sum_ranks_kernel(const char *slots, size_t cap, int world, char *result, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        T acc = ((const T *)slots)[i];
        for (int k = 1; k < world; ++k) acc += ((const T *)(slots + (size_t)k * cap))[i];      // rank order: deterministic
        ((T *)result)[i] = acc;
    }
}

// rank 0, between two host barriers: grows the staging sets (rare: sizes repeat) -- the only place that waits for the device, and
// only for the GROUP's stream plus the recorded reader events of every set, never for a caller's stream
bool ensure_capacity(group *g, size_t bytes)
{
    if (bytes <= g->cap) return true;
    for (int s = 0; s < GENERATIONS; ++s) {
        for (int k = 0; k < g->world; ++k) if (hipEventSynchronize(g->ev_done[s][k]) != hipSuccess) return false;
        if (hipEventSynchronize(g->ev_out[s]) != hipSuccess) return false;
    }
    const size_t cap = (bytes + 4095) & ~(size_t)4095;
    for (int s = 0; s < GENERATIONS; ++s) {
        if (g->slots[s]) (void)hipFree(g->slots[s]);
        if (g->result[s]) (void)hipFree(g->result[s]);
        if (hipMalloc((void **)&g->slots[s], cap * g->world) != hipSuccess || hipMalloc((void **)&g->result[s], cap) != hipSuccess) return false;
    }
    g->cap = cap;
    return true;
}

// common head of every collective: agree on the size (mismatched calls fail on EVERY rank), pick the staging set, make this rank's
// stream wait for the previous readers of its slot in that set
ncclResult_t begin(comm *c, size_t bytes, hipStream_t st, int &set)
{
    group *g = c->g;
    g->posted[c->rank] = bytes;
    g->barrier();
    if (c->rank == 0) {
        g->err = 0;
        for (int k = 0; k < g->world; ++k) if (g->posted[k] != bytes) g->err = 1;
        if (!g->err && !ensure_capacity(g, bytes)) g->err = 2;
    }
    g->barrier();
    if (g->err) return g->err == 1 ? ncclInvalidArgument : ncclUnhandledCudaError;
    set = (int)(g->seq % GENERATIONS);
    // this set's previous collective: its sum has read my slot (ev_out) and every rank may still be copying out of the slots /
    // the result (ev_done): both must be over before my stream overwrites my slot
    if (hipStreamWaitEvent(st, g->ev_out[set], 0) != hipSuccess) return ncclUnhandledCudaError;
    for (int k = 0; k < g->world; ++k) if (hipStreamWaitEvent(st, g->ev_done[set][k], 0) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
void finish(comm *c)
{
    group *g = c->g;
    g->barrier();                                          // every rank has enqueued its part
    if (c->rank == 0) ++g->seq;
    g->barrier();
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
    if (memcmp(id.internal + 8, "LOOPBACK", 8) != 0 || nranks < 1 || nranks > MAX_WORLD || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    group *g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_groups.find(v);
        if (it == g_groups.end()) {
            g = new group; g->world = nranks;
            bool ok = true;
            for (int s = 0; s < GENERATIONS && ok; ++s) {
                ok = hipStreamCreateWithFlags(&g->gs[s], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&g->ev_out[s], hipEventDisableTiming) == hipSuccess;
                for (int k = 0; k < nranks && ok; ++k)
                    ok = hipEventCreateWithFlags(&g->ev_in[s][k], hipEventDisableTiming) == hipSuccess &&
                         hipEventCreateWithFlags(&g->ev_done[s][k], hipEventDisableTiming) == hipSuccess;
            }
            if (!ok) { delete g; return ncclUnhandledCudaError; }
            g_groups[v] = g;
        } else g = it->second;
        if (g->world != nranks) return ncclInvalidArgument;
    }
    g->barrier();                                       // collective like the real one: returns once every rank has joined
    comm *c = new comm{g, rank};
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t cm) { delete (comm *)cm; return ncclSuccess; }   // the group's few device buffers live as long as the process (tests)
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "loopback collective error"; }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t cm, hipStream_t st)
{
    comm *c = (comm *)cm; group *g = c->g;
    const size_t bytes = count * dsize(dt);
    if (op != ncclSum || bytes == 0) return ncclInvalidArgument;
    int set = 0;
    ncclResult_t r = begin(c, bytes, st, set);
    if (r != ncclSuccess) return r;
    bool ok = hipMemcpyAsync(g->slots[set] + (size_t)c->rank * g->cap, send, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess &&
              hipEventRecord(g->ev_in[set][c->rank], st) == hipSuccess;
    g->barrier();                                       // all contributions are enqueued and their events recorded
    if (c->rank == 0 && ok) {
        for (int k = 0; k < g->world; ++k) ok = ok && hipStreamWaitEvent(g->gs[set], g->ev_in[set][k], 0) == hipSuccess;
        const int blocks = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
        static const long delay_us = getenv("LOOPBACK_DELAY_US") ? atol(getenv("LOOPBACK_DELAY_US")) : 0;
        static const long delay_min = getenv("LOOPBACK_DELAY_MIN_COUNT") ? atol(getenv("LOOPBACK_DELAY_MIN_COUNT")) : 0;   // only collectives of at least that many elements are late
        if (delay_us > 0 && (long)count >= delay_min) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, g->gs[set], (long long)delay_us * 100);
        switch (dt) {
        case ncclFloat32: hipLaunchKernelGGL(sum_ranks_kernel<float>, dim3(blocks), dim3(256), 0, g->gs[set], g->slots[set], g->cap, g->world, g->result[set], count); break;
        case ncclFloat64: hipLaunchKernelGGL(sum_ranks_kernel<double>, dim3(blocks), dim3(256), 0, g->gs[set], g->slots[set], g->cap, g->world, g->result[set], count); break;
        case ncclInt32: hipLaunchKernelGGL(sum_ranks_kernel<int>, dim3(blocks), dim3(256), 0, g->gs[set], g->slots[set], g->cap, g->world, g->result[set], count); break;
        case ncclUint32: hipLaunchKernelGGL(sum_ranks_kernel<unsigned>, dim3(blocks), dim3(256), 0, g->gs[set], g->slots[set], g->cap, g->world, g->result[set], count); break;
        default: ok = false;
        }
        ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(g->ev_out[set], g->gs[set]) == hipSuccess;
        if (!ok) g->err = 2;
    }
    g->barrier();                                       // the sum is enqueued, ev_out recorded
    ok = ok && !g->err && hipStreamWaitEvent(st, g->ev_out[set], 0) == hipSuccess &&
         hipMemcpyAsync(recv, g->result[set], bytes, hipMemcpyDeviceToDevice, st) == hipSuccess &&
         hipEventRecord(g->ev_done[set][c->rank], st) == hipSuccess;
    finish(c);
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t cm, hipStream_t st)
{
    comm *c = (comm *)cm; group *g = c->g;
    const size_t bytes = sendcount * dsize(dt);
    if (bytes == 0) return ncclInvalidArgument;
    int set = 0;
    ncclResult_t r = begin(c, bytes, st, set);
    if (r != ncclSuccess) return r;
    bool ok = hipMemcpyAsync(g->slots[set] + (size_t)c->rank * g->cap, send, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess &&
              hipEventRecord(g->ev_in[set][c->rank], st) == hipSuccess;
    g->barrier();
    for (int k = 0; k < g->world && ok; ++k)
        ok = hipStreamWaitEvent(st, g->ev_in[set][k], 0) == hipSuccess &&
             hipMemcpyAsync((char *)recv + (size_t)k * bytes, g->slots[set] + (size_t)k * g->cap, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    ok = ok && hipEventRecord(g->ev_done[set][c->rank], st) == hipSuccess;
    finish(c);
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t cm, hipStream_t st)
{
    comm *c = (comm *)cm; group *g = c->g;
    const size_t bytes = count * dsize(dt);
    if (bytes == 0 || root < 0 || root >= g->world) return ncclInvalidArgument;
    int set = 0;
    ncclResult_t r = begin(c, bytes, st, set);
    if (r != ncclSuccess) return r;
    bool ok = true;
    if (c->rank == root)
        ok = hipMemcpyAsync(g->slots[set] + (size_t)root * g->cap, send, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess &&
             hipEventRecord(g->ev_in[set][root], st) == hipSuccess;
    g->barrier();
    ok = ok && hipStreamWaitEvent(st, g->ev_in[set][root], 0) == hipSuccess &&
         hipMemcpyAsync(recv, g->slots[set] + (size_t)root * g->cap, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess &&
         hipEventRecord(g->ev_done[set][c->rank], st) == hipSuccess;
    finish(c);
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}
}
