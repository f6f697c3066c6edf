// multi.cpp -- native multi-GPU host path (multi.h): shard, launch side by side, gather the objectives over RCCL.
#include "multi.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <exception>
#include <cstring>
#include <stdexcept>

#include "kernels.h"
#include "knobs.h"

namespace tolfg {

namespace {

void check(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
        throw hip_failure(std::string(what) + ": " + hipGetErrorString(e));
    }
}

// the calling thread's current device is the caller's business: put it back whatever this library selected meanwhile
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// rccl.h: ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0 (/opt/rocm/include/rccl/rccl.h:448-467)
enum { kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0 };

void clear_hip_errors()
{
    for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
}

void nccl_check(int rc, const char *what)
{
    if (rc != 0) throw hip_failure(std::string(what) + ": " + rccl_api::get().GetErrorString(rc));
}

// directory of the HIP runtime this process has mapped: its librccl is the one that matches it
std::string hip_runtime_dir()
{
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t a = p.find_last_of('/');
        if (a != std::string::npos) return p.substr(0, a + 1);
    }
    return std::string();
}

}  // namespace

void shard_bounds(long total, int rank, int world, long *lo, long *hi)
{
    const long base = total / world, extra = total % world;
    *lo = rank * base + (rank < extra ? rank : extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

long shard_width(long total, int world)
{
    long lo, hi;
    shard_bounds(total, 0, world, &lo, &hi);
    return hi - lo;
}

void compact_gathered(const void *padded, size_t elem, long total, int world, void *out)
{
    const long width = shard_width(total, world);
    for (int r = 0; r < world; ++r) {
        long lo, hi;
        shard_bounds(total, r, world, &lo, &hi);
        std::memcpy(static_cast<char *>(out) + elem * (size_t)lo, static_cast<const char *>(padded) + elem * (size_t)(r * width),
                    elem * (size_t)(hi - lo));
    }
}

const rccl_api &rccl_api::get()
{
    static rccl_api api;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [] {
        std::vector<std::string> names;
        const std::string dir = hip_runtime_dir();
        const std::string &named = knobs().rccl_library;          // knobs.h
        if (!named.empty()) {
            // an explicit choice is final: that library or a failure, never a silent second pick
            api.handle = dlopen(named.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (!api.handle) {
                const char *why = dlerror();
                failure = "the collective library named in the environment, " + named + ", cannot be loaded: " + (why ? why : "?");
                return;
            }
            api.path = named;
        } else {
            if (!dir.empty()) { names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
            names.emplace_back("librccl.so.1");
            names.emplace_back("librccl.so");
            // a copy the process already holds wins (PyTorch maps its own); otherwise the one beside the HIP runtime
            for (int pass = 0; pass < 2 && !api.handle; ++pass)
                for (const std::string &n : names) {
                    api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                    if (api.handle) { api.path = n; break; }
                }
        }
        if (!api.handle) {
            failure = "librccl was not found (tried the HIP runtime's directory '" + dir + "' and the default search path; include/tolfg.h, \"Environment\", says how to name one)";
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(api.handle, name);
            if (!p && failure.empty()) failure = std::string("librccl (") + api.path + ") has no symbol " + name;
            return p;
        };
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    });
    if (!failure.empty()) throw hip_failure(failure);
    return api;
}

multi::multi(const std::string &mission, const std::string &root, const std::vector<std::string> &names, int ts, int windmodel,
             int dtype, int pattern, const std::vector<int> &devices)
    : dev_(devices), dtype_(dtype)
{
    if (devices.empty() || devices.size() > 64) throw std::invalid_argument("tolfg_multi: 1..64 devices");
    // Test seam (knobs.h): with the shared-devices variable AND an explicit stand-in collective library an ordinal may appear
    // more than once, so that several parts -- their issuing threads, shard dealing, per-shard uploads and the padded gather --
    // run on a box with ONE GPU (tests/loopback_nccl).  RCCL itself refuses duplicate devices in ncclCommInitAll.
    if (!knobs().multi_shared_devices && !knobs().multi_solo_comms)
        for (size_t i = 0; i < devices.size(); ++i)
            for (size_t j = 0; j < i; ++j)
                if (devices[i] == devices[j]) throw std::invalid_argument("tolfg_multi: every device may appear once");
    part_.resize(devices.size());
    for (size_t i = 0; i < devices.size(); ++i) {
        Part &p = part_[i];
        p.device = devices[i];
        p.b.reset(new batch(mission, root, names, ts, windmodel, dtype, devices[i], pattern));   // host-side set-up only
    }
    // device state: needs the GPUs (no CPU path)
    DeviceGuard guard;
    try {
        int ndev = 0;
        check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        for (int d : devices)
            if (d < 0 || d >= ndev) throw std::invalid_argument("tolfg_multi: no such HIP device");
        for (size_t i = 0; i < part_.size(); ++i) {
            Part &p = part_[i];
            p.index = (int)i;
            check(hipSetDevice(p.device), "hipSetDevice");
            check(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking), "hipStreamCreate");
            // the collective's kernels on a high-priority stream: a gather queued beside a launch that has every CU booked gets
            // onto the chip when the first wave slots free up, not when the launch's backlog of tiles is through
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; clear_hip_errors(); }
            check(hipStreamCreateWithPriority(&p.gstream, hipStreamNonBlocking, knobs().multi_gather_priority ? greatest : least), "hipStreamCreateWithPriority");
            for (int k = 0; k < kSlots; ++k) {
                check(hipEventCreateWithFlags(&p.ev_launch[k], hipEventDisableTiming), "hipEventCreate");
                check(hipEventCreateWithFlags(&p.ev_gather[k], hipEventDisableTiming), "hipEventCreate");
            }
            check(hipEventCreate(&p.t0), "hipEventCreate");
            check(hipEventCreate(&p.t1), "hipEventCreate");
            check(hipMalloc(&p.dSum, 2 * sizeof(double)), "hipMalloc(sum)");
        }
        const rccl_api &nc = rccl_api::get();
        std::vector<void *> comms(devices.size(), nullptr);
        if (knobs().multi_solo_comms) {
            // measurement build only (knobs.h): every part a communicator of ONE rank, so that several parts can sit on one device
            // under the real library -- the gathered vectors then hold only the part's own block: for timing the host side, nothing else
            for (size_t i = 0; i < devices.size(); ++i) nccl_check(nc.CommInitAll(&comms[i], 1, &devices[i]), "ncclCommInitAll(solo)");
        } else {
            nccl_check(nc.CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
        }
        for (size_t i = 0; i < devices.size(); ++i) part_[i].comm = comms[i];
        for (size_t i = 1; i < devices.size(); ++i) threads_.emplace_back(&multi::worker, this, (int)i);
    } catch (...) {
        release();
        throw;
    }
}

multi::~multi() { release(); }

void multi::release()
{
    DeviceGuard guard;
    {
        std::lock_guard<std::mutex> lk(mu_);
        quit_ = true;
        ++generation_;
        gen_hint_.store(generation_, std::memory_order_release);
    }
    cv_go_.notify_all();
    for (std::thread &t : threads_) t.join();
    threads_.clear();
    for (Part &p : part_) {
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        if (p.gstream) (void)hipStreamSynchronize(p.gstream);
    }
    for (Part &p : part_)
        if (p.comm) { (void)rccl_api::get().CommDestroy(p.comm); p.comm = nullptr; }      // a comm exists only if the api loaded
    free_buffers();
    for (Part &p : part_) {
        (void)hipSetDevice(p.device);
        if (p.dSum) (void)hipFree(p.dSum);
        p.dSum = nullptr;
        p.b.reset();
        for (int k = 0; k < kSlots; ++k) {
            if (p.ev_launch[k]) (void)hipEventDestroy(p.ev_launch[k]);
            if (p.ev_gather[k]) (void)hipEventDestroy(p.ev_gather[k]);
            p.ev_launch[k] = p.ev_gather[k] = nullptr;
        }
        if (p.t0) (void)hipEventDestroy(p.t0);
        if (p.t1) (void)hipEventDestroy(p.t1);
        p.t0 = p.t1 = nullptr;
        if (p.gstream) (void)hipStreamDestroy(p.gstream);
        if (p.stream) (void)hipStreamDestroy(p.stream);
        p.stream = p.gstream = nullptr;
    }
    clear_hip_errors();
}

void multi::free_buffers()
{
    DeviceGuard guard;
    for (int k = 0; k < kSlots; ++k) {
        if (hAll_[k]) (void)hipHostFree(hAll_[k]);
        if (hObj_[k]) (void)hipHostFree(hObj_[k]);
        hAll_[k] = hObj_[k] = nullptr;
    }
    for (Part &p : part_) {
        (void)hipSetDevice(p.device);
        for (void **q : {&p.dX, &p.dF, &p.dWind}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        for (int k = 0; k < kSlots; ++k)
            for (void **q : {&p.dObj[k], &p.dAll[k]}) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
        if (p.dG) {
            try { device_free(p.dG); } catch (const std::exception &) {}
            p.dG = nullptr;
        }
    }
}

namespace {
// A short spin before a condition-variable sleep: in a step loop the next job (or the last worker's completion) is a few
// microseconds away, and a futex sleep + wake costs 10-20 us of latency per hand-over (measured: 3 us per step with one part, 16-26
// with worker threads in play, profiles/r05_native_multi.md).  Bounded: 30 us, then the thread sleeps like before.
template <typename Pred>
bool spin_until(Pred &&ready)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    for (int i = 0;; ++i) {
        if (ready()) return true;
        __builtin_ia32_pause();
        if ((i & 63) == 63 && clk::now() - t0 > std::chrono::microseconds(30)) return false;
    }
}
}  // namespace

void multi::worker(int i)
{
    unsigned long seen = 0;
    (void)hipSetDevice(part_[i].device);
    for (;;) {
        const std::function<void(Part &)> *job = nullptr;
        (void)spin_until([&] { return gen_hint_.load(std::memory_order_acquire) != seen; });
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_go_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            if (quit_) return;
            job = job_;
        }
        std::string err;
        try {
            (*job)(part_[i]);
        } catch (const std::exception &e) {
            err = e.what();
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!err.empty()) errors_.push_back("device " + std::to_string(part_[i].device) + ": " + err);
            --pending_;
            pending_hint_.store(pending_, std::memory_order_release);
        }
        cv_done_.notify_one();
    }
}

void multi::on_every_device(const std::function<void(Part &)> &fn)
{
    DeviceGuard guard;
    {
        std::lock_guard<std::mutex> lk(mu_);
        job_ = &fn;
        pending_ = (int)threads_.size();
        pending_hint_.store(pending_, std::memory_order_relaxed);
        errors_.clear();
        ++generation_;
        gen_hint_.store(generation_, std::memory_order_release);
    }
    cv_go_.notify_all();
    std::string mine;
    try {
        check(hipSetDevice(part_[0].device), "hipSetDevice");
        fn(part_[0]);
    } catch (const std::exception &e) {
        mine = std::string("device ") + std::to_string(part_[0].device) + ": " + e.what();
    }
    (void)spin_until([&] { return pending_hint_.load(std::memory_order_acquire) == 0; });
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    if (!mine.empty()) errors_.insert(errors_.begin(), mine);
    if (!errors_.empty()) {
        std::string all;
        for (const std::string &e : errors_) all += (all.empty() ? "" : "; ") + e;
        throw hip_failure(all);
    }
}

const void *multi::gathered(int i) const
{
    return gather_ == GATHER_HOST ? hObj_[last_gather_slot_] : part_.at(i).dAll[last_gather_slot_];
}

int multi::nccl_type() const { return dtype_ == TOLFG_F64 ? kNcclFloat64 : kNcclFloat32; }

int multi::rccl_version() const
{
    int v = 0;
    const rccl_api &nc = rccl_api::get();
    return (nc.GetVersion && nc.GetVersion(&v) == 0) ? v : 0;
}

void multi::set_issue(int mode)
{
    if (mode != ISSUE_GROUPED && mode != ISSUE_THREADS) throw std::invalid_argument("tolfg_multi_set_issue: unknown mode");
    issue_ = mode;
}

void multi::set_gather(int mode)
{
    if (mode != GATHER_RCCL && mode != GATHER_HOST) throw std::invalid_argument("tolfg_multi_set_gather: unknown mode");
    sync();
    gather_ = mode;
}

void multi::set_trajectories(long total, const tolfg_traj *trajs, int place_tries)
{
    if (total < 1 || !trajs) throw std::invalid_argument("tolfg_multi_set_trajectories: at least one trajectory and a table");
    sync();
    const int world = devices();
    const Sizes &sz = sizes();
    const long v = dtype_ == TOLFG_F64 ? 2 : 4;
    auto up = [&](long m) { return (m + v - 1) / v * v; };
    free_buffers();
    total_ = total;
    width_ = shard_width(total, world);
    ldx_ = up(sz.n); ldf_ = up(sz.neF); ldg_ = up(sz.neG);
    seq_ = 0;
    last_gather_slot_ = 0;
    for (int i = 0; i < world; ++i) shard_bounds(total, i, world, &part_[i].lo, &part_[i].hi);
    {
        DeviceGuard guard;
        check(hipSetDevice(part_[0].device), "hipSetDevice");
        for (int k = 0; k < kSlots; ++k) {
            check(hipHostMalloc(&hAll_[k], elem() * (size_t)width_ * world, hipHostMallocDefault), "hipHostMalloc(gathered)");
            // portable + mapped: one host vector every device can store into (GATHER_HOST); a pinned host pointer is its own
            // device address under unified addressing, on every device
            check(hipHostMalloc(&hObj_[k], elem() * (size_t)total, hipHostMallocPortable | hipHostMallocMapped), "hipHostMalloc(host objectives)");
            std::memset(hObj_[k], 0, elem() * (size_t)total);
        }
    }
    on_every_device([&](Part &p) {
        check(hipSetDevice(p.device), "hipSetDevice");
        const long B = p.hi - p.lo;
        if (B > 0) p.b->set_trajectories((int)B, trajs + p.lo);
        const size_t rows = (size_t)(B > 0 ? B : 1);
        check(hipMalloc(&p.dX, elem() * rows * ldx_), "hipMalloc(X)");
        check(hipMalloc(&p.dF, elem() * rows * ldf_), "hipMalloc(F)");
        // G, the bulk of what a launch writes, comes placed for this shard's launch (problem.h: alloc_outputs): up to
        // place_tries candidates per device, held side by side within half of its free memory, ~0.2 s each at 1.4 GB, the devices
        // searching concurrently on their own threads
        if (B > 0) {
            long ldg = 0;
            p.dG = p.b->alloc_outputs((int)B, place_tries, &ldg, nullptr, nullptr);
            if (ldg != ldg_) throw std::logic_error("tolfg_multi: row stride of the placed G buffer");
        } else {
            p.dG = device_alloc(p.device, elem() * rows * ldg_);
        }
        // everything below is ordered on the launch stream, ahead of the first evaluation
        for (int k = 0; k < kSlots; ++k) {
            check(hipMalloc(&p.dObj[k], elem() * (size_t)width_), "hipMalloc(obj)");
            check(hipMalloc(&p.dAll[k], elem() * (size_t)width_ * world), "hipMalloc(gathered)");
            check(hipMemsetAsync(p.dObj[k], 0, elem() * (size_t)width_, p.stream), "hipMemsetAsync(obj)");
        }
        check(hipMemsetAsync(p.dX, 0, elem() * rows * ldx_, p.stream), "hipMemsetAsync(X)");
    });
}

void multi::buffers(int i, void **dX, long *ldx, void **dF, long *ldf, void **dG, long *ldg) const
{
    const Part &p = part_.at(i);
    if (dX) *dX = p.dX;
    if (dF) *dF = p.dF;
    if (dG) *dG = p.dG;
    if (ldx) *ldx = ldx_;
    if (ldf) *ldf = ldf_;
    if (ldg) *ldg = ldg_;
}

void multi::x0()
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    on_every_device([&](Part &p) {
        if (p.hi > p.lo) p.b->x0_device((int)(p.hi - p.lo), p.dX, ldx_, p.stream);
    });
}

void multi::set_wind_grid(const tolfg_wind_grid &g)
{
    sync();
    on_every_device([&](Part &p) { p.b->set_wind_grid(g); });
}

void multi::set_wind_tables(const double *wind_enu)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    if (!wind_enu) throw std::invalid_argument("tolfg_multi_set_wind_tables: null table");
    sync();
    const size_t per = 12 * (size_t)(sizes().N + 1);
    on_every_device([&](Part &p) {
        const size_t rows = (size_t)(p.hi - p.lo);
        if (rows == 0) return;
        check(hipSetDevice(p.device), "hipSetDevice");
        if (!p.dWind) check(hipMalloc(&p.dWind, elem() * rows * per), "hipMalloc(wind)");
        // from pinned staging ON THE LAUNCH STREAM, which is then drained: the copy is ordered ahead of the evaluations
        // that read it whatever the null stream does, and the staging can go
        const double *src = wind_enu + (size_t)p.lo * per;
        void *stage = nullptr;
        check(hipHostMalloc(&stage, elem() * rows * per, hipHostMallocDefault), "hipHostMalloc(wind staging)");
        if (dtype_ == TOLFG_F64) std::memcpy(stage, src, sizeof(double) * rows * per);
        else { float *f = static_cast<float *>(stage); for (size_t i = 0; i < rows * per; ++i) f[i] = (float)src[i]; }
        hipError_t e = hipMemcpyAsync(p.dWind, stage, elem() * rows * per, hipMemcpyHostToDevice, p.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p.stream);
        (void)hipHostFree(stage);
        check(e, "upload(wind tables)");
    });
}

// ---- the pieces of a step (the part's device is current)

void multi::part_eval(Part &p, int slot, bool needF, bool needG, const void *X)
{
    if (p.hi <= p.lo) return;
    if (needF && gather_ == GATHER_RCCL) {
        // (GATHER_HOST: the event of kSlots steps back sits on this very stream, ahead of this launch)
        // slot reuse: the gather that last read this objective buffer (kSlots gathers back) must be through.  It normally is,
        // long ago -- then nothing is put into the launch stream (a wait marker between two launches costs the second a few us)
        const hipError_t q = hipEventQuery(p.ev_gather[slot]);
        if (q == hipErrorNotReady) {
            clear_hip_errors();
            // The issuing thread is kSlots steps ahead of its device.  It waits HERE, on the host, for that gather: the launch
            // stream still holds kSlots - 1 launches to run meanwhile, and gets no wait marker (which would cost every launch of a
            // host that runs ahead -- every host does -- a few microseconds: profiles/r05_native_multi.md).  It also bounds how far
            // the host runs ahead of the devices.
            if (knobs().multi_slot_wait_on_host) check(hipEventSynchronize(p.ev_gather[slot]), "hipEventSynchronize(slot reuse)");
            else check(hipStreamWaitEvent(p.stream, p.ev_gather[slot], 0), "hipStreamWaitEvent(slot reuse)");
        } else if (q != hipSuccess) {
            check(q, "hipEventQuery");
        }
    }
    // where the finalizing waves put the objectives: the device buffer the all-gather sends, or (GATHER_HOST) this shard's
    // place in the pinned host vector
    void *obj = gather_ == GATHER_HOST ? static_cast<char *>(hObj_[slot]) + elem() * (size_t)p.lo : p.dObj[slot];
    p.b->eval((int)(p.hi - p.lo), X ? X : p.dX, ldx_, p.dF, ldf_, p.dG, ldg_, p.dWind, needF ? 1 : 0, needG ? 1 : 0, p.stream,
              needF ? obj : nullptr);
}

void multi::part_gather_pre(Part &p, int slot)
{
    if (gather_ == GATHER_HOST) {      // the whole gather: an event behind the launch whose waves wrote the host vector
        check(hipEventRecord(p.ev_gather[slot], p.stream), "hipEventRecord(host gather)");
        return;
    }
    // the gather stream picks up where the launch stream stands now: behind the evaluation whose objectives it carries
    check(hipEventRecord(p.ev_launch[slot], p.stream), "hipEventRecord(launch)");
    check(hipStreamWaitEvent(p.gstream, p.ev_launch[slot], 0), "hipStreamWaitEvent(gather)");
}

void multi::part_gather_call(Part &p, int slot)
{
    if (gather_ == GATHER_HOST) return;
    nccl_check(rccl_api::get().AllGather(p.dObj[slot], p.dAll[slot], (size_t)width_, nccl_type(), p.comm, p.gstream), "ncclAllGather");
}

void multi::part_gather_post(Part &p, int slot)
{
    if (gather_ == GATHER_HOST) return;
    check(hipEventRecord(p.ev_gather[slot], p.gstream), "hipEventRecord(gather)");
}

void multi::eval(bool needF, bool needG, const void *const *dX)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    const int slot = (int)(seq_ % kSlots);
    on_every_device([&](Part &p) { part_eval(p, slot, needF, needG, dX ? dX[p.index] : nullptr); });
    evaluated_since_gather_ = true;
}

unsigned long multi::gather_begin()
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    const int slot = (int)(seq_ % kSlots);
    if (issue_ == ISSUE_THREADS) {
        on_every_device([&](Part &p) { part_gather_pre(p, slot); part_gather_call(p, slot); part_gather_post(p, slot); });
    } else {
        // one group call from this thread: the single-process form of a collective over several devices
        DeviceGuard guard;
        const rccl_api &nc = rccl_api::get();
        for (Part &p : part_) { check(hipSetDevice(p.device), "hipSetDevice"); part_gather_pre(p, slot); }
        if (gather_ == GATHER_RCCL) {
            nccl_check(nc.GroupStart(), "ncclGroupStart");
            for (Part &p : part_) part_gather_call(p, slot);
            nccl_check(nc.GroupEnd(), "ncclGroupEnd");
            for (Part &p : part_) { check(hipSetDevice(p.device), "hipSetDevice"); part_gather_post(p, slot); }
        }
    }
    last_gather_slot_ = slot;
    evaluated_since_gather_ = false;
    return seq_++;
}

unsigned long multi::step(bool needF, bool needG, const void *const *dX)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    if (!needF) throw std::invalid_argument("tolfg_multi_step: the gather carries the objectives, so F is needed");
    if (issue_ != ISSUE_THREADS && gather_ == GATHER_RCCL) {
        eval(needF, needG, dX);
        return gather_begin();
    }
    // one wake-up of the issuing threads per step: every thread issues its launch and its part of the gather
    const int slot = (int)(seq_ % kSlots);
    on_every_device([&](Part &p) {
        // a device whose launch fails still makes its call of the collective: the other devices' calls are on their way, and
        // a collective one rank never joins would leave their kernels waiting for ever
        std::exception_ptr failed;
        try { part_eval(p, slot, needF, needG, dX ? dX[p.index] : nullptr); } catch (...) { failed = std::current_exception(); }
        part_gather_pre(p, slot); part_gather_call(p, slot); part_gather_post(p, slot);
        if (failed) std::rethrow_exception(failed);
    });
    last_gather_slot_ = slot;
    evaluated_since_gather_ = false;
    return seq_++;
}

void multi::gather_wait(unsigned long ticket, void *host_out)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    if (ticket >= seq_ || seq_ - ticket > (unsigned long)kSlots)
        throw std::invalid_argument("tolfg_multi_gather_wait: no such gather in flight (a ticket stays valid for " + std::to_string(kSlots) +
                                    " further gathers)");
    const int slot = (int)(ticket % kSlots);
    {
        DeviceGuard guard;
        for (Part &p : part_) {
            check(hipSetDevice(p.device), "hipSetDevice");
            check(hipEventSynchronize(p.ev_gather[slot]), "hipEventSynchronize(gather)");
        }
    }
    // the gather is behind the evaluation that fed it: that evaluation's health can be asked now
    for (Part &p : part_)
        if (p.b->take_lost_partial()) throw hip_failure("device " + std::to_string(p.device) + ": an evaluation lost an objective partial");
    if (host_out && gather_ == GATHER_HOST) {
        std::memcpy(host_out, hObj_[slot], elem() * (size_t)total_);      // global order already: every shard was stored at its offset
    } else if (host_out) {
        // device 0's copy of the gathered vector comes to the host only when somebody asks for it: a step that nobody reads
        // on the host (the steady state of a Monte-Carlo loop) carries no copy command
        Part &p0 = part_[0];
        DeviceGuard guard;
        check(hipSetDevice(p0.device), "hipSetDevice");
        check(hipMemcpyAsync(hAll_[slot], p0.dAll[slot], elem() * (size_t)width_ * devices(), hipMemcpyDeviceToHost, p0.gstream),
              "hipMemcpyAsync(gathered)");
        check(hipStreamSynchronize(p0.gstream), "hipStreamSynchronize(gathered)");
        compact_gathered(hAll_[slot], elem(), total_, devices(), host_out);
    }
}

void multi::gather_objectives(void *host_out)
{
    const unsigned long t = gather_begin();
    gather_wait(t, host_out);
    sync();
}

double multi::mean_objective()
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    if (gather_ == GATHER_HOST) {      // the objectives are on the host already: wait for the launches, add them up in global order
        sync();
        const int hs = (int)((seq_ > 0 && !evaluated_since_gather_ ? seq_ - 1 : seq_) % kSlots);
        double sum = 0.0;
        for (long t = 0; t < total_; ++t) sum += dtype_ == TOLFG_F64 ? static_cast<const double *>(hObj_[hs])[t] : (double)static_cast<const float *>(hObj_[hs])[t];
        return sum / (double)total_;
    }
    const rccl_api &nc = rccl_api::get();
    // the objectives of the evaluation issued last: the slot the next gather would carry, or -- right after a gather, nothing
    // evaluated since -- the one it carried; either way the buffer the launch stream wrote last
    const int slot = (int)((seq_ > 0 && !evaluated_since_gather_ ? seq_ - 1 : seq_) % kSlots);
    on_every_device([&](Part &p) {
        check(launch_sum(p.dObj[slot], (int)(p.hi - p.lo), dtype_, static_cast<double *>(p.dSum), p.stream), "launch sum");
    });
    nccl_check(nc.GroupStart(), "ncclGroupStart");
    for (Part &p : part_)
        nccl_check(nc.AllReduce(p.dSum, static_cast<double *>(p.dSum) + 1, 1, kNcclFloat64, kNcclSum, p.comm, p.stream), "ncclAllReduce");
    nccl_check(nc.GroupEnd(), "ncclGroupEnd");
    sync();
    double s[2];
    DeviceGuard guard;
    check(hipSetDevice(part_[0].device), "hipSetDevice");
    check(hipMemcpy(s, part_[0].dSum, sizeof s, hipMemcpyDeviceToHost), "hipMemcpy(sum)");      // device to host: complete on return
    return s[1] / (double)total_;
}

void multi::sync()
{
    DeviceGuard guard;
    for (Part &p : part_) {
        check(hipSetDevice(p.device), "hipSetDevice");
        check(hipStreamSynchronize(p.stream), "hipStreamSynchronize");
        check(hipStreamSynchronize(p.gstream), "hipStreamSynchronize(gather)");
    }
}

// ---- the native step loop (measurement aid)

void multi::steps_run(int n, int n_x, const void *const *dX, bool needF, bool needG, bool gather, unsigned long first_step)
{
    const int world = devices();
    if (gather && (issue_ == ISSUE_THREADS || gather_ == GATHER_HOST)) {
        // every device's thread issues its own n steps -- launch, event, wait, its communicator's all-gather, copy, event --
        // without meeting the other threads on the host: the collective's kernels meet on the devices
        const unsigned long seq0 = seq_;
        on_every_device([&](Part &p) {
            std::exception_ptr failed;      // a device that fails keeps joining the collectives of the remaining steps (see step())
            for (int i = 0; i < n; ++i) {
                const int slot = (int)((seq0 + (unsigned long)i) % kSlots);
                const void *X = n_x > 0 ? dX[(size_t)((first_step + (unsigned long)i) % (unsigned long)n_x) * world + p.index] : nullptr;
                if (!failed) {
                    try { part_eval(p, slot, needF, needG, X); } catch (...) { failed = std::current_exception(); }
                }
                part_gather_pre(p, slot); part_gather_call(p, slot); part_gather_post(p, slot);
            }
            if (failed) std::rethrow_exception(failed);
        });
        seq_ += (unsigned long)n;
        if (n > 0) { last_gather_slot_ = (int)((seq_ - 1) % kSlots); evaluated_since_gather_ = false; }
        return;
    }
    for (int i = 0; i < n; ++i) {
        const void *const *X = n_x > 0 ? dX + (size_t)((first_step + (unsigned long)i) % (unsigned long)n_x) * world : nullptr;
        if (gather) step(needF, needG, X);
        else eval(needF, needG, X);
    }
}

multi::Timing multi::time_steps(int n_x, const void *const *dX, bool needF, bool needG, bool gather, int warm, int steps,
                                double *launch_us_per_device)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    if (steps < 1 || warm < 0 || n_x < 0 || (n_x > 0 && !dX)) throw std::invalid_argument("tolfg_multi_time_steps: bad arguments");
    if (gather && !needF) throw std::invalid_argument("tolfg_multi_time_steps: the gather carries the objectives, so F is needed");
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    Timing t;
    steps_run(warm, n_x, dX, needF, needG, gather, 0);
    sync();
    on_every_device([&](Part &p) { check(hipEventRecord(p.t0, p.stream), "hipEventRecord"); });
    const auto h0 = clk::now();
    steps_run(steps, n_x, dX, needF, needG, gather, (unsigned long)warm);
    const auto h1 = clk::now();
    on_every_device([&](Part &p) { check(hipEventRecord(p.t1, p.stream), "hipEventRecord"); });
    if (gather) gather_wait(seq_ - 1, nullptr);
    sync();
    const auto h2 = clk::now();
    t.wall_us_per_step = us(h0, h2) / steps;
    t.issue_us_per_step = us(h0, h1) / steps;
    {
        DeviceGuard guard;
        for (Part &p : part_) {
            check(hipSetDevice(p.device), "hipSetDevice");
            float ms = 0;
            check(hipEventElapsedTime(&ms, p.t0, p.t1), "hipEventElapsedTime");
            const double per = p.hi > p.lo ? 1e3 * ms / steps : 0.0;
            if (launch_us_per_device) launch_us_per_device[p.index] = per;
            if (per > t.launch_us_per_step) t.launch_us_per_step = per;
        }
    }
    if (gather) {      // the gather alone: synchronous, nothing else in flight
        const int reps = 50;
        for (int i = 0; i < 5; ++i) gather_wait(gather_begin(), nullptr);
        sync();
        const auto g0 = clk::now();
        for (int i = 0; i < reps; ++i) gather_wait(gather_begin(), nullptr);
        t.gather_us = us(g0, clk::now()) / reps;
    }
    return t;
}

}  // namespace tolfg
