// multi.cpp -- native multi-GPU host path (multi.h): shard, launch side by side, gather the objectives over RCCL.
#include "multi.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "kernels.h"

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
        if (const char *e = std::getenv("TOLFG_RCCL_LIBRARY")) {
            // an explicit choice is final: that library or a failure, never a silent second pick
            api.handle = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
            if (!api.handle) {
                const char *why = dlerror();
                failure = std::string("TOLFG_RCCL_LIBRARY=") + e + " cannot be loaded: " + (why ? why : "?");
                return;
            }
            api.path = e;
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
            failure = "librccl was not found (tried the HIP runtime's directory '" + dir + "' and the default search path; TOLFG_RCCL_LIBRARY names one explicitly)";
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
    // Test seam: TOLFG_MULTI_SHARED_DEVICES=1 lets an ordinal appear more than once, so that several parts -- their
    // issuing threads, shard dealing, per-shard uploads and the padded gather -- run on a box with ONE GPU.  RCCL itself
    // refuses duplicate devices in ncclCommInitAll, so this only works with a stand-in collective library named by
    // TOLFG_RCCL_LIBRARY (tests/loopback_nccl).  Never set in production.
    const char *shared = std::getenv("TOLFG_MULTI_SHARED_DEVICES");
    if (!(shared && shared[0] == '1'))
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
        for (Part &p : part_) {
            check(hipSetDevice(p.device), "hipSetDevice");
            check(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking), "hipStreamCreate");
            check(hipMalloc(&p.dSum, 2 * sizeof(double)), "hipMalloc(sum)");
        }
        const rccl_api &nc = rccl_api::get();
        std::vector<void *> comms(devices.size(), nullptr);
        nccl_check(nc.CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
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
    }
    cv_go_.notify_all();
    for (std::thread &t : threads_) t.join();
    threads_.clear();
    for (Part &p : part_) {
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
    }
    for (Part &p : part_)
        if (p.comm) { (void)rccl_api::get().CommDestroy(p.comm); p.comm = nullptr; }      // a comm exists only if the api loaded
    free_buffers();
    for (Part &p : part_) {
        (void)hipSetDevice(p.device);
        if (p.dSum) (void)hipFree(p.dSum);
        p.dSum = nullptr;
        p.b.reset();
        if (p.stream) (void)hipStreamDestroy(p.stream);
        p.stream = nullptr;
    }
    for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
}

void multi::free_buffers()
{
    DeviceGuard guard;
    if (hAll_) (void)hipHostFree(hAll_);
    hAll_ = nullptr;
    for (Part &p : part_) {
        (void)hipSetDevice(p.device);
        for (void **q : {&p.dX, &p.dF, &p.dObj, &p.dAll, &p.dWind}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        if (p.dG) {
            try { device_free(p.dG); } catch (const std::exception &) {}
            p.dG = nullptr;
        }
    }
}

void multi::worker(int i)
{
    unsigned long seen = 0;
    (void)hipSetDevice(part_[i].device);
    for (;;) {
        const std::function<void(Part &)> *job = nullptr;
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
        errors_.clear();
        ++generation_;
    }
    cv_go_.notify_all();
    std::string mine;
    try {
        check(hipSetDevice(part_[0].device), "hipSetDevice");
        fn(part_[0]);
    } catch (const std::exception &e) {
        mine = std::string("device ") + std::to_string(part_[0].device) + ": " + e.what();
    }
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    if (!mine.empty()) errors_.insert(errors_.begin(), mine);
    if (!errors_.empty()) {
        std::string all;
        for (const std::string &e : errors_) all += (all.empty() ? "" : "; ") + e;
        throw hip_failure(all);
    }
}

void multi::set_trajectories(long total, const tolfg_traj *trajs)
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
    for (int i = 0; i < world; ++i) shard_bounds(total, i, world, &part_[i].lo, &part_[i].hi);
    {
        DeviceGuard guard;
        check(hipSetDevice(part_[0].device), "hipSetDevice");
        check(hipHostMalloc(&hAll_, elem() * (size_t)width_ * world, hipHostMallocDefault), "hipHostMalloc(gathered)");
    }
    on_every_device([&](Part &p) {
        check(hipSetDevice(p.device), "hipSetDevice");
        const long B = p.hi - p.lo;
        if (B > 0) p.b->set_trajectories((int)B, trajs + p.lo);
        const size_t rows = (size_t)(B > 0 ? B : 1);
        check(hipMalloc(&p.dX, elem() * rows * ldx_), "hipMalloc(X)");
        check(hipMalloc(&p.dF, elem() * rows * ldf_), "hipMalloc(F)");
        // G, the bulk of what a launch writes, comes placed for this shard's launch (problem.h: alloc_outputs)
        if (B > 0) {
            long ldg = 0;
            p.dG = p.b->alloc_outputs((int)B, 12, &ldg, nullptr, nullptr);
            if (ldg != ldg_) throw std::logic_error("tolfg_multi: row stride of the placed G buffer");
        } else {
            p.dG = device_alloc(p.device, elem() * rows * ldg_);
        }
        check(hipMalloc(&p.dObj, elem() * (size_t)width_), "hipMalloc(obj)");
        check(hipMalloc(&p.dAll, elem() * (size_t)width_ * world), "hipMalloc(gathered)");
        check(hipMemsetAsync(p.dObj, 0, elem() * (size_t)width_, p.stream), "hipMemsetAsync(obj)");
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
        const double *src = wind_enu + (size_t)p.lo * per;
        if (dtype_ == TOLFG_F64) {
            check(hipMemcpy(p.dWind, src, sizeof(double) * rows * per, hipMemcpyHostToDevice), "hipMemcpy(wind)");
        } else {
            std::vector<float> tmp(src, src + rows * per);
            check(hipMemcpy(p.dWind, tmp.data(), sizeof(float) * rows * per, hipMemcpyHostToDevice), "hipMemcpy(wind)");
        }
    });
}

void multi::eval(bool needF, bool needG)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    on_every_device([&](Part &p) {
        if (p.hi > p.lo)
            p.b->eval((int)(p.hi - p.lo), p.dX, ldx_, p.dF, ldf_, p.dG, ldg_, p.dWind, needF ? 1 : 0, needG ? 1 : 0, p.stream,
                      needF ? p.dObj : nullptr);
    });
}

void multi::gather_objectives(void *host_out)
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    const rccl_api &nc = rccl_api::get();
    // one group call from this thread: the single-process form of a collective over several devices
    nccl_check(nc.GroupStart(), "ncclGroupStart");
    for (Part &p : part_)
        nccl_check(nc.AllGather(p.dObj, p.dAll, (size_t)width_, dtype_ == TOLFG_F64 ? kNcclFloat64 : kNcclFloat32, p.comm, p.stream),
                   "ncclAllGather");
    nccl_check(nc.GroupEnd(), "ncclGroupEnd");
    if (host_out) {      // device 0's copy of the gathered vector follows its gather on the same stream, into pinned memory
        DeviceGuard guard;
        check(hipSetDevice(part_[0].device), "hipSetDevice");
        check(hipMemcpyAsync(hAll_, part_[0].dAll, elem() * (size_t)width_ * devices(), hipMemcpyDeviceToHost, part_[0].stream),
              "hipMemcpyAsync(gathered)");
    }
    sync();
    for (Part &p : part_)
        if (p.b->take_lost_partial()) throw hip_failure("device " + std::to_string(p.device) + ": an evaluation lost an objective partial");
    if (host_out) compact_gathered(hAll_, elem(), total_, devices(), host_out);
}

double multi::mean_objective()
{
    if (total_ < 1) throw std::invalid_argument("tolfg_multi: set_trajectories first");
    const rccl_api &nc = rccl_api::get();
    on_every_device([&](Part &p) {
        check(launch_sum(p.dObj, (int)(p.hi - p.lo), dtype_, static_cast<double *>(p.dSum), p.stream), "launch sum");
    });
    nccl_check(nc.GroupStart(), "ncclGroupStart");
    for (Part &p : part_)
        nccl_check(nc.AllReduce(p.dSum, static_cast<double *>(p.dSum) + 1, 1, kNcclFloat64, kNcclSum, p.comm, p.stream), "ncclAllReduce");
    nccl_check(nc.GroupEnd(), "ncclGroupEnd");
    sync();
    double s[2];
    DeviceGuard guard;
    check(hipSetDevice(part_[0].device), "hipSetDevice");
    check(hipMemcpy(s, part_[0].dSum, sizeof s, hipMemcpyDeviceToHost), "hipMemcpy(sum)");
    return s[1] / (double)total_;
}

void multi::sync()
{
    DeviceGuard guard;
    for (Part &p : part_) {
        check(hipSetDevice(p.device), "hipSetDevice");
        check(hipStreamSynchronize(p.stream), "hipStreamSynchronize");
    }
}

}  // namespace tolfg
