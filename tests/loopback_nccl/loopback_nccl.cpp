// loopback_nccl.cpp -- TEST DOUBLE, not product code: the eight nccl* entry points tol_amd/csrc/multi.cpp resolves at
// run time (multi.h: rccl_api), implemented with device-to-device copies between the ranks' buffers inside ONE process.
//
// Why it exists: this pool's test boxes have one GPU, and RCCL's ncclCommInitAll refuses a device list that names a
// device twice.  With TOLFG_RCCL_LIBRARY=<this library> and TOLFG_MULTI_SHARED_DEVICES=1 the native multi-GPU host path
// (worker threads, shard dealing, per-shard uploads, the padded all-gather, the all-reduce of the partial sums) runs
// with several parts on one device (tests/test_multi_loopback.py).  It implements the collectives' CONTRACT -- every
// rank's receive buffer holds every rank's send block in rank order; stream-ordered after each rank's earlier work --
// not their transport: nothing here says anything about RCCL over xGMI.
//
// Semantics kept from NCCL: calls between ncclGroupStart / ncclGroupEnd are deferred to the outermost ncclGroupEnd; a
// collective runs once every rank of its communicator set has posted its call; the result is ordered on each rank's
// stream after the work that stream already held, and the send buffers may be reused by work queued afterwards.
// (Unlike NCCL it blocks the host while the streams drain: see run().)
#include <hip/hip_runtime.h>

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
    std::vector<Comm *> comm;
    std::vector<Op> op;
    std::vector<hipEvent_t> before, after;
};

std::mutex mu;
thread_local int group_depth = 0;
thread_local std::vector<World *> deferred;

size_t elem_size(int dtype) { return dtype == 7 ? 4 : dtype == 8 ? 8 : 0; }      // ncclFloat32 = 7, ncclFloat64 = 8

#define HIP_OK(call) do { if ((call) != hipSuccess) return kUnhandledHipError; } while (0)

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
    // 2. the data movement.  The streams are drained on the host first: without that, a small device-to-device
    //    hipMemcpyAsync queued right behind the evaluation kernel on the same stream was seen (2 runs in 25, ROCm 7.2.0) to
    //    deliver 0.0 for the objective the kernel's LAST finalizing wave writes -- a stale read this test double has no
    //    business depending on either way.  RCCL's collectives are kernels; a test double can afford to block.
    for (int s = 0; s < n; ++s) {
        HIP_OK(hipSetDevice(w->comm[s]->device));
        HIP_OK(hipStreamSynchronize(w->op[s].stream));
    }
    if (w->op[0].kind == kAllGather) {
        for (int r = 0; r < n; ++r) {
            HIP_OK(hipSetDevice(w->comm[r]->device));
            for (int s = 0; s < n; ++s)
                HIP_OK(hipMemcpyAsync(static_cast<char *>(w->op[r].recv) + bytes * s, w->op[s].send, bytes, hipMemcpyDeviceToDevice, w->op[r].stream));
        }
    } else {      // sum, in rank order, on the host (a test double: a few doubles)
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
    for (Op &o : w->op) o = Op();
    w->posted = 0;
    HIP_OK(hipSetDevice(prev));
    return kSuccess;
}

int post(void *comm, const Op &op)
{
    if (!comm || !op.send || !op.recv || elem_size(op.dtype) == 0) return kInvalidArgument;
    Comm *c = static_cast<Comm *>(comm);
    World *w = c->world;
    std::lock_guard<std::mutex> lk(mu);
    if (w->op[c->rank].kind != 0) return kInvalidUsage;          // a rank posts once per collective
    w->op[c->rank] = op;
    if (++w->posted < w->n) return kSuccess;
    if (group_depth > 0) { deferred.push_back(w); return kSuccess; }
    return run(w);
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
        const int one = run(w);
        if (rc == kSuccess) rc = one;
    }
    deferred.clear();
    return rc;
}

}  // extern "C"
