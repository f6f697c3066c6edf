// loopback_nccl.cpp -- TEST DOUBLE, not product code: the eight nccl* entry points tol_amd/csrc/multi.cpp resolves at
// run time (multi.h: rccl_api), implemented with copies between the ranks' buffers inside ONE process.
//
// Why it exists: this pool's test boxes have one GPU, and RCCL's ncclCommInitAll refuses a device list that names a
// device twice.  With TOLFG_RCCL_LIBRARY=<this library> and TOLFG_MULTI_SHARED_DEVICES=1 the native multi-GPU host path
// (worker threads, shard dealing, per-shard uploads, the padded all-gather on the gather streams, the all-reduce of the
// partial sums) runs with several parts on one device (tests/test_multi_loopback.py).  It implements the collectives'
// CONTRACT -- every rank's receive buffer holds every rank's send block in rank order; stream-ordered after each rank's
// earlier work; asynchronous to the host -- not their transport: nothing here says anything about RCCL over xGMI.
//
// Semantics kept from NCCL:
//  * calls between ncclGroupStart / ncclGroupEnd are deferred to the outermost ncclGroupEnd;
//  * a call outside a group, one thread per rank: it returns once the collective is enqueued on its stream, which needs
//    every rank's call -- so it blocks until the last rank has posted (as NCCL's may), and one thread posting for several
//    ranks outside a group would wait for ever (NCCL: use a group);
//  * the all-gather is ASYNCHRONOUS to the host: copy kernels enqueued on each rank's stream behind events on the
//    other ranks' streams (the producers of the send buffers), and no stream runs ahead of the collective (work queued
//    after it may overwrite the send buffers).  Ranks on one device copy with a kernel, like RCCL does; ranks on different
//    devices fall back to hipMemcpyAsync.
//  * the all-reduce (a few doubles, synchronous use only) blocks and sums on the host in rank order.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

enum { kSuccess = 0, kUnhandledHipError = 1, kInvalidArgument = 4, kInvalidUsage = 5 };
enum { kAllGather = 1, kAllReduce = 2 };

struct World;
struct Comm {
    World *world;
    int rank, device;
};
struct Op {
    int kind = 0;
    const void *send = nullptr;
    void *recv = nullptr;
    size_t count = 0;
    int dtype = 0;
    hipStream_t stream = nullptr;
};
struct World {
    int n = 0, alive = 0, posted = 0;
    unsigned long round = 0;             // collectives run so far
    int last_rc = kSuccess;
    std::vector<Comm *> comm;
    std::vector<Op> op;
    std::vector<hipEvent_t> before, after;
};

std::mutex mu;
std::condition_variable cv;
thread_local int group_depth = 0;
thread_local std::vector<World *> deferred;

size_t elem_size(int dtype) { return dtype == 7 ? 4 : dtype == 8 ? 8 : 0; }      // ncclFloat32 = 7, ncclFloat64 = 8

#define HIP_OK(call) do { if ((call) != hipSuccess) return kUnhandledHipError; } while (0)

__global__ void copy_kernel(const unsigned *__restrict__ src, unsigned *__restrict__ dst, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int run(World *w)
{
    const int n = w->n;
    int prev = 0;
    HIP_OK(hipGetDevice(&prev));
    for (int r = 1; r < n; ++r)
        if (w->op[r].kind != w->op[0].kind || w->op[r].count != w->op[0].count || w->op[r].dtype != w->op[0].dtype) return kInvalidUsage;
    const size_t bytes = w->op[0].count * elem_size(w->op[0].dtype);
    // 1. what each stream already holds (the producers of the send buffers) completes before any copy reads them
    for (int s = 0; s < n; ++s) {
        HIP_OK(hipSetDevice(w->comm[s]->device));
        HIP_OK(hipEventRecord(w->before[s], w->op[s].stream));
    }
    for (int r = 0; r < n; ++r) {
        HIP_OK(hipSetDevice(w->comm[r]->device));
        for (int s = 0; s < n; ++s)
            if (s != r) HIP_OK(hipStreamWaitEvent(w->op[r].stream, w->before[s], 0));
    }
    // 2. the data movement
    if (w->op[0].kind == kAllGather) {
        for (int r = 0; r < n; ++r) {
            HIP_OK(hipSetDevice(w->comm[r]->device));
            for (int s = 0; s < n; ++s) {
                char *dst = static_cast<char *>(w->op[r].recv) + bytes * s;
                if (w->comm[s]->device == w->comm[r]->device && bytes % 4 == 0) {
                    const size_t words = bytes / 4;
                    const unsigned blocks = (unsigned)((words + 255) / 256 > 64 ? 64 : (words + 255) / 256);
                    hipLaunchKernelGGL(copy_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, w->op[r].stream,
                                       static_cast<const unsigned *>(w->op[s].send), reinterpret_cast<unsigned *>(dst), words);
                    HIP_OK(hipGetLastError());
                } else {
                    HIP_OK(hipMemcpyAsync(dst, w->op[s].send, bytes, hipMemcpyDeviceToDevice, w->op[r].stream));
                }
            }
        }
    } else {      // sum, in rank order, on the host (a test double: a few doubles; blocks)
        for (int s = 0; s < n; ++s) {
            HIP_OK(hipSetDevice(w->comm[s]->device));
            HIP_OK(hipStreamSynchronize(w->op[s].stream));
        }
        std::vector<char> acc(bytes, 0), one(bytes);
        for (int s = 0; s < n; ++s) {
            HIP_OK(hipSetDevice(w->comm[s]->device));
            HIP_OK(hipMemcpy(one.data(), w->op[s].send, bytes, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < w->op[0].count; ++i) {
                if (w->op[0].dtype == 8) reinterpret_cast<double *>(acc.data())[i] += reinterpret_cast<const double *>(one.data())[i];
                else reinterpret_cast<float *>(acc.data())[i] += reinterpret_cast<const float *>(one.data())[i];
            }
        }
        for (int r = 0; r < n; ++r) {
            HIP_OK(hipSetDevice(w->comm[r]->device));
            HIP_OK(hipMemcpy(w->op[r].recv, acc.data(), bytes, hipMemcpyHostToDevice));
            HIP_OK(hipDeviceSynchronize());
        }
    }
    // 3. no stream runs ahead of the collective: work queued after it may overwrite the send buffers
    for (int r = 0; r < n; ++r) {
        HIP_OK(hipSetDevice(w->comm[r]->device));
        HIP_OK(hipEventRecord(w->after[r], w->op[r].stream));
    }
    for (int s = 0; s < n; ++s) {
        HIP_OK(hipSetDevice(w->comm[s]->device));
        for (int r = 0; r < n; ++r)
            if (r != s) HIP_OK(hipStreamWaitEvent(w->op[s].stream, w->after[r], 0));
    }
    HIP_OK(hipSetDevice(prev));
    return kSuccess;
}

// with `mu` held: run the collective every rank has posted, wake the ranks that wait for it
int run_posted(World *w)
{
    const int rc = run(w);
    for (Op &o : w->op) o = Op();
    w->posted = 0;
    w->last_rc = rc;
    ++w->round;
    cv.notify_all();
    return rc;
}

int post(void *comm, const Op &op)
{
    if (!comm || !op.send || !op.recv || elem_size(op.dtype) == 0) return kInvalidArgument;
    Comm *c = static_cast<Comm *>(comm);
    World *w = c->world;
    std::unique_lock<std::mutex> lk(mu);
    if (w->op[c->rank].kind != 0) return kInvalidUsage;          // a rank posts once per collective
    w->op[c->rank] = op;
    const bool last = ++w->posted == w->n;
    if (group_depth > 0) {                                       // inside a group: nothing happens before ncclGroupEnd
        if (last) deferred.push_back(w);
        return kSuccess;
    }
    if (last) return run_posted(w);
    // one thread per rank, no group: the call returns when the collective is on this rank's stream
    const unsigned long mine = w->round;
    cv.wait(lk, [&] { return w->round != mine; });
    return w->last_rc;
}

}  // namespace

extern "C" {

int ncclGetVersion(int *version) { if (version) *version = 0; return kSuccess; }

const char *ncclGetErrorString(int rc)
{
    switch (rc) {
    case kSuccess: return "no error (loop-back test double)";
    case kUnhandledHipError: return "loop-back test double: a HIP call failed";
    case kInvalidArgument: return "loop-back test double: invalid argument";
    case kInvalidUsage: return "loop-back test double: invalid usage (mismatched or repeated calls of one collective)";
    default: return "loop-back test double: unknown error";
    }
}

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return kInvalidArgument;
    int prev = 0;
    HIP_OK(hipGetDevice(&prev));
    World *w = new World;
    w->n = w->alive = ndev;
    w->op.resize(ndev);
    w->before.resize(ndev);
    w->after.resize(ndev);
    for (int r = 0; r < ndev; ++r) {
        Comm *c = new Comm{w, r, devlist ? devlist[r] : r};
        w->comm.push_back(c);
        comms[r] = c;
        HIP_OK(hipSetDevice(c->device));
        HIP_OK(hipEventCreateWithFlags(&w->before[r], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&w->after[r], hipEventDisableTiming));
    }
    HIP_OK(hipSetDevice(prev));
    return kSuccess;
}

int ncclCommDestroy(void *comm)
{
    if (!comm) return kInvalidArgument;
    Comm *c = static_cast<Comm *>(comm);
    World *w = c->world;
    std::lock_guard<std::mutex> lk(mu);
    (void)hipEventDestroy(w->before[c->rank]);
    (void)hipEventDestroy(w->after[c->rank]);
    delete c;
    if (--w->alive == 0) delete w;
    return kSuccess;
}

int ncclAllGather(const void *send, void *recv, size_t sendcount, int datatype, void *comm, hipStream_t stream)
{
    Op op;
    op.kind = kAllGather; op.send = send; op.recv = recv; op.count = sendcount; op.dtype = datatype; op.stream = stream;
    return post(comm, op);
}

int ncclAllReduce(const void *send, void *recv, size_t count, int datatype, int redop, void *comm, hipStream_t stream)
{
    if (redop != 0) return kInvalidArgument;          // ncclSum only
    Op op;
    op.kind = kAllReduce; op.send = send; op.recv = recv; op.count = count; op.dtype = datatype; op.stream = stream;
    return post(comm, op);
}

int ncclGroupStart() { ++group_depth; return kSuccess; }

int ncclGroupEnd()
{
    if (group_depth <= 0) return kInvalidUsage;
    if (--group_depth > 0) return kSuccess;
    std::lock_guard<std::mutex> lk(mu);
    int rc = kSuccess;
    for (World *w : deferred) {
        const int one = run_posted(w);
        if (rc == kSuccess) rc = one;
    }
    deferred.clear();
    return rc;
}

}  // extern "C"
