// problem.cpp -- host side: batched evaluator and the SNOPT-facing problem objects.
#include "problem.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "knobs.h"

namespace tolfg {

problem *prob = nullptr;

namespace {

constexpr double kRho = 1.2682;   // ref: include/problem.h:73

void check(hipError_t e, const char *what)
{
    if (e != hipSuccess)
        throw hip_failure(std::string(what) + ": " + hipGetErrorString(e));
}

// Drop error codes left behind by calls whose failure was handled (a bounded loop: without a device
// the runtime reports its absence for ever).
void clear_errors()
{
    for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
}

// Set-up uploads that are not tied to a launch stream (wind grid, wind table: the caller has no evaluation in flight, as the
// header asks).  hipMemcpy from pageable memory may return once the source is staged, before the DMA has landed, and the
// null stream it rides is not ordered against the non-blocking streams the evaluations use: so the device is drained
// before this returns, and whatever stream the next evaluation names finds the data in place.
void blocking_upload(void *dst, const void *src, size_t bytes, const char *what)
{
    check(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), what);
    check(hipDeviceSynchronize(), what);
}

int kernel_wind(int windmodel)
{
    switch (windmodel) {
    case TOLFG_WIND_NONE:  return WIND_NONE;
    case TOLFG_WIND_SHEAR: return WIND_SHEAR;
    case TOLFG_WIND_TABLE: return WIND_TABLE;
    case TOLFG_WIND_GRID:  return WIND_GRID;
    // the reference's thermal (2), two-thermal (4) and cyclic (5) arms have their bodies commented out (src/problem.cpp:534-542,
    // 698-730): modelWind leaves the wind vectors as the constructor zeroed them, i.e. no wind
    case 2: case 4: case 5: return WIND_NONE;
    }
    throw std::invalid_argument("unknown wind model");
}

}  // namespace

int mission_from_name(const std::string &name, bool allow_mixed)
{
    if (name == "S10") return MISSION_S10;
    if (name == "G7") return MISSION_G7;
    if (allow_mixed && (name == "mixed" || name == "S10+G7" || name == "G7+S10")) return MISSION_MIXED;
    throw std::invalid_argument("Mission code \"" + name + "\" not recognized.");   // ref: src/tol.cpp:21
}

std::string default_root()
{
    // <repo>/tol_amd/lib/libtolfg.so -> <repo>/tol_amd/data/
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&default_root), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t a = p.find_last_of('/');
        if (a != std::string::npos) {
            p.erase(a);
            const size_t b = p.find_last_of('/');
            if (b != std::string::npos) return p.substr(0, b) + "/data/";
        }
    }
    return "./";   // the reference's command-line root_path (src/arguments.cpp:45)
}

// ------------------------------------------------------------------------------------ placed device memory

namespace {
// the calling thread's current device is the caller's business
struct DeviceGuardLocal {
    int prev = -1;
    DeviceGuardLocal() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuardLocal() { if (prev >= 0) (void)hipSetDevice(prev); }
};
struct DeviceBlock {
    int device = 0;
    size_t bytes = 0, chunk = 0;                           // chunk = 0: a plain hipMalloc (no virtual-memory support)
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::mutex g_blocks_mu;
std::map<void *, DeviceBlock> g_blocks;
void note_freed(int device, size_t bytes);       // settle_block's bookkeeping, below

void release_block(void *ptr, DeviceBlock &b)
{
    (void)hipSetDevice(b.device);
    if (b.chunk == 0) { (void)hipFree(ptr); return; }
    // hipFree waits for the device by itself; unmapping does not, and a launch still writing the range would fault
    (void)hipDeviceSynchronize();
    // every chunk unmapped the way it was mapped, one mapping at a time (one hipMemUnmap over the whole range leans on the runtime
    // splitting it over the mappings it covers)
    for (size_t i = 0; i < b.handles.size(); ++i) (void)hipMemUnmap(static_cast<char *>(ptr) + i * b.chunk, b.chunk);
    for (hipMemGenericAllocationHandle_t h : b.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(ptr, b.bytes);
    note_freed(b.device, b.bytes);          // the driver wipes what it gets back, later: the next device_alloc waits accordingly (settle_block)
}
}  // namespace

namespace {

// A block that hipMemCreate has just produced is NOT yet the caller's to write.  The driver wipes VRAM it gets back with a copy-engine
// job of its own, hands the chunks out again before that job has run, and neither hipMemMap nor hipMemSetAccess waits for it: measured
// on MI355X / ROCm 7.2 (tools/placed_fresh_write.py, profiles/r05_fresh_vmm_blocks.md), a kernel that fills an 8 MB block right after
// device_alloc finds up to whole 2 MiB chunks back at 0.0 a few milliseconds later in ~50 % of allocations that follow a free -- silently.
// Zeroing the block first does not help, waiting for the chunks' fences through a dma-buf poll does not either, blocks allocated
// without a free before them are (all but) safe, and hipMalloc never shows it.  So the allocator settles a block before it hands it
// out: the block is filled with a pattern and must still hold it, every word, through a quiet period -- what a copy engine at a
// conservative 10 GB/s needs for this block plus everything this library freed on the device during the last second, at least 1 ms; a word that
// went back to zero restarts the wait.  450 back-to-back allocations after frees: none lost a write (against 40-90 % without).
// Cost: one fill and a few reads of the block, and the quiet period: ~1.5 ms for a small block, ~0.1 s per GB: set-up time.
constexpr unsigned kSettlePattern = 0xA5C35A3Cu;
using settle_clock = std::chrono::steady_clock;
std::mutex g_freed_mu;
std::map<int, std::vector<std::pair<settle_clock::time_point, size_t>>> g_freed;       // per device: what device_free gave back to the driver lately

void note_freed(int device, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_freed_mu);
    auto &v = g_freed[device];
    const auto now = settle_clock::now();
    v.emplace_back(now, bytes);
    while (!v.empty() && now - v.front().first > std::chrono::seconds(1)) v.erase(v.begin());
}

size_t freed_lately(int device)
{
    std::lock_guard<std::mutex> lk(g_freed_mu);
    const auto it = g_freed.find(device);
    if (it == g_freed.end()) return 0;
    const auto now = settle_clock::now();
    size_t sum = 0;
    for (const auto &f : it->second)
        if (now - f.first <= std::chrono::seconds(1)) sum += f.second;
    return sum;
}

void settle_block(void *ptr, const DeviceBlock &b)
{
    unsigned long long *count = nullptr;
    check(hipHostMalloc(reinterpret_cast<void **>(&count), sizeof *count, hipHostMallocDefault), "hipHostMalloc(settle)");
    struct Free { void *p; ~Free() { (void)hipHostFree(p); } } guard{count};
    const double quiet_ms = std::min(2000.0, std::max(1.0, (double)(b.bytes + freed_lately(b.device)) / 10e6));
    const auto nap = std::chrono::microseconds((long)std::min(5000.0, std::max(200.0, 1e3 * quiet_ms / 20)));
    const auto deadline = settle_clock::now() + std::chrono::milliseconds(3000 + (long)(4 * quiet_ms));
    for (;;) {
        check(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ptr), (int)kSettlePattern, b.bytes / 4, nullptr), "hipMemsetD32Async(settle)");
        check(hipDeviceSynchronize(), "hipDeviceSynchronize(settle)");
        const auto filled = settle_clock::now();
        bool held = true;
        for (;;) {
            *count = 0;
            check(launch_count_not(ptr, b.bytes, kSettlePattern, count, nullptr), "launch count");
            check(hipDeviceSynchronize(), "hipDeviceSynchronize(settle)");
            if (*count != 0) { held = false; break; }
            if (std::chrono::duration<double, std::milli>(settle_clock::now() - filled).count() >= quiet_ms) break;
            std::this_thread::sleep_for(nap);
        }
        if (held) {       // handed out zeroed, as the driver's own clear leaves a fresh block
            check(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ptr), 0, b.bytes / 4, nullptr), "hipMemsetD32Async(settle)");
            check(hipDeviceSynchronize(), "hipDeviceSynchronize(settle)");
            return;
        }
        if (settle_clock::now() > deadline) throw hip_failure("device_alloc: a fresh block kept losing what was written to it (the driver's wipe did not settle)");
    }
}

}  // namespace

void *device_alloc(int device, size_t bytes)
{
    if (bytes == 0) throw std::invalid_argument("device_alloc: zero bytes");
    const size_t kPlacedChunk = knobs().placed_chunk;       // 2 MiB (knobs.h: the measurement build can change it)
    DeviceGuardLocal guard;
    check(hipSetDevice(device), "hipSetDevice");
    DeviceBlock blk;
    blk.device = device;
    void *ptr = nullptr;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    const bool settle = knobs().place_settle != 0;    // always, but for the measurement build's A/B (settle_block above)
    size_t gran = 0;
    bool vmm = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0 &&
               kPlacedChunk % gran == 0;
    if (vmm) {
        blk.chunk = kPlacedChunk;
        blk.bytes = (bytes + kPlacedChunk - 1) / kPlacedChunk * kPlacedChunk;
        if (hipMemAddressReserve(&ptr, blk.bytes, kPlacedChunk, nullptr, 0) != hipSuccess || !ptr) { vmm = false; ptr = nullptr; }
    }
    if (vmm) {
        bool ok = true;
        size_t mapped = 0;
        for (size_t off = 0; off < blk.bytes && ok; off += kPlacedChunk) {
            hipMemGenericAllocationHandle_t h{};
            ok = hipMemCreate(&h, kPlacedChunk, &prop, 0) == hipSuccess;
            if (!ok) break;
            blk.handles.push_back(h);
            ok = hipMemMap(static_cast<char *>(ptr) + off, kPlacedChunk, 0, h, 0) == hipSuccess;
            if (ok) mapped = off + kPlacedChunk;
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (ok) ok = hipMemSetAccess(ptr, blk.bytes, &acc, 1) == hipSuccess;
        std::string why = "the virtual-memory allocation of " + std::to_string(bytes) + " bytes failed";
        if (ok && settle) {
            try { settle_block(ptr, blk); } catch (const std::exception &e) { ok = false; why = e.what(); }
        }
        if (!ok) {             // out of memory or an unsupported step: undo, report
            for (size_t off = 0; off < mapped; off += kPlacedChunk) (void)hipMemUnmap(static_cast<char *>(ptr) + off, kPlacedChunk);
            for (hipMemGenericAllocationHandle_t h : blk.handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(ptr, blk.bytes);
            clear_errors();
            throw hip_failure("device_alloc: " + why);
        }
    } else {
        clear_errors();
        blk = DeviceBlock{};
        blk.device = device;
        blk.bytes = bytes;
        check(hipMalloc(&ptr, bytes), "hipMalloc");
    }
    std::lock_guard<std::mutex> lk(g_blocks_mu);
    g_blocks[ptr] = std::move(blk);
    return ptr;
}

void device_free(void *ptr)
{
    if (!ptr) return;
    DeviceBlock blk;
    {
        std::lock_guard<std::mutex> lk(g_blocks_mu);
        auto it = g_blocks.find(ptr);
        if (it == g_blocks.end()) throw std::invalid_argument("device_free: not a pointer tolfg_device_alloc returned");
        blk = std::move(it->second);
        g_blocks.erase(it);
    }
    DeviceGuardLocal guard;
    release_block(ptr, blk);
}

// ------------------------------------------------------------------------------------ batch

batch::batch(const std::string &mission, const std::string &root, const std::vector<std::string> &names,
             int ts, int windmodel, int dtype, int device, int pattern)
    : windmodel_(windmodel), dtype_(dtype), device_(device)
{
    mission_ = mission_from_name(mission, true);
    if (names.empty() || names.size() > MAX_AIRCRAFT) throw std::invalid_argument("1..8 aircraft per batch");
    if (dtype != TOLFG_F64 && dtype != TOLFG_F32) throw std::invalid_argument("dtype");
    kernel_wind(windmodel);
    if (windmodel == 2 || windmodel == 4 || windmodel == 5) windmodel_ = TOLFG_WIND_NONE;
    for (const std::string &nm : names) acs_.emplace_back(nm, root);
    const char *mname[2] = {"S10", "G7"};
    for (int m = 0; m < 2; ++m) {
        if (mission_ != MISSION_MIXED && m != mission_) continue;
        gn_.emplace_back(mname[m], root); lm_.emplace_back(mname[m], root); sn_.emplace_back(mname[m], root);
        const snopt &sn = sn_.back();
        if (sn.numinp != 11 || sn.numstates != 8 || sn.numbounds != (m == MISSION_S10 ? 11 : 12))
            throw std::invalid_argument(std::string("snopt.param: numinp/numstates/numbounds do not describe ") + mname[m]);
    }
    const int N = ts > 0 ? ts : sn_.front().ts;        // a mixed batch shares ts (S10's snopt.param when not given)
    if (N < 1) throw std::invalid_argument("ts must be >= 1");
    if (pattern != PATTERN_REFERENCE && pattern != PATTERN_COMPACT) throw std::invalid_argument("pattern");
    for (int m = 0; m < 2; ++m) szm_[m] = make_sizes(m, N, pattern);
    if (mission_ == MISSION_MIXED) {
        sz_ = szm_[MISSION_S10];
        sz_.mission = MISSION_MIXED;
        sz_.nb = std::max(szm_[0].nb, szm_[1].nb);
        sz_.neF = std::max(szm_[0].neF, szm_[1].neF);
        sz_.neG = std::max(szm_[0].neG, szm_[1].neG);
    } else {
        sz_ = szm_[mission_];
    }

    args_.pattern = pattern;
    // overrides of the launch plan (plan.cpp): none in the shipped library; the measurement build takes them from the
    // environment (knobs.h), clamped to what launch_fg accepts: a tile size is a multiple of 4 in [4, 64] (fp32: up to 128)
    refresh_knobs();
    const Knobs &kn = knobs();
    auto tile_ok = [&](int v) { return v <= 0 ? 0 : std::min(std::max(v & ~3, 4), dtype == TOLFG_F32 ? 2 * kTileNodes : kTileNodes); };
    if (kn.waves_per_cu >= 0) { waves_per_cu_ = kn.waves_per_cu; waves_forced_ = true; }
    args_.N = N;
    plan_tiles(N, dtype, 0, &args_.tiles, &args_.nt);      // eval() re-plans for its batch size
    if (kn.tile_nodes >= 0) tile_nodes_forced_ = tile_ok(kn.tile_nodes);
    fused_forced_ = kn.fused; nt_forced_ = kn.nt_stores; xcd_forced_ = kn.xcd; stagger_forced_ = kn.stagger; sub_forced_ = kn.sub_nodes;
    tail_forced_ = kn.tail_count;
    if (kn.tail_nt > 0) tail_nt_forced_ = std::min(tile_ok(kn.tail_nt), kTileNodes);
    no_single_ = kn.no_single_launch; force_single_ = kn.force_single_launch; x0_serial_ = kn.x0_serial;
    for (int m = 0; m < 2; ++m) {
        const gain &g = gains(m);
        args_.c0[m] = szm_[m].c0;
        args_.kT[m] = g.kT; args_.kp[m] = g.kp; args_.kv[m] = g.kv; args_.kdt[m] = g.kdt;
    }
    for (size_t i = 0; i < acs_.size(); ++i) {
        const aircraft &a = acs_[i];
        args_.ac[i].inv_m = 1.0 / a.mm;
        args_.ac[i].qk = kRho * a.SS / (2.0 * a.mm);
        args_.ac[i].Cd0 = a.Cd0;
        args_.ac[i].kind = 1.0 / (a.AR * M_PI * a.ee);
        args_.ac[i].mm = a.mm; args_.ac[i].SS = a.SS; args_.ac[i].AR = a.AR; args_.ac[i].ee = a.ee;
    }
}

const Sizes &batch::sizes_of(int mission) const
{
    if (mission != MISSION_S10 && mission != MISSION_G7) throw std::invalid_argument("mission id");
    if (mission_ != MISSION_MIXED && mission != mission_) throw std::invalid_argument("this batch holds one mission only");
    return szm_[mission];
}

int batch::mission_of_traj(int t) const
{
    return mission_ == MISSION_MIXED ? host_traj_.at(t).mission : mission_;
}

double batch::algorithmic_bytes(int B) const
{
    double elems = 0;
    for (int t = 0; t < B; ++t) {
        const Sizes &z = (mission_ == MISSION_MIXED && t < ntraj_) ? szm_[host_traj_[t].mission] : (mission_ == MISSION_MIXED ? szm_[0] : sz_);
        elems += (double)z.n + z.neF + z.neG;
    }
    return elems * (double)elem_size();
}

batch::~batch()
{
    if (d_traj_) (void)hipFree(d_traj_);
    if (d_partial_) (void)hipFree(d_partial_);
    if (d_counter_) (void)hipFree(d_counter_);
    if (h_status_) (void)hipHostFree(h_status_);
    if (d_grid_) (void)hipFree(d_grid_);
    if (d_tgrid_) (void)hipFree(d_tgrid_);
    for (hipEvent_t e : ev_) if (e) (void)hipEventDestroy(e);
}

void batch::set_wind_grid(const tolfg_wind_grid &g)
{
    if (!g.v || g.nx < 2 || g.ny < 2 || g.nz < 2 || !(g.dx > 0) || !(g.dy > 0) || !(g.dz > 0))
        throw std::invalid_argument("wind grid: need >= 2 points per axis, positive spacing and values");
    const size_t cnt = (size_t)g.nx * g.ny * g.nz;
    check(hipSetDevice(device_), "hipSetDevice");
    if (d_grid_) check(hipFree(d_grid_), "hipFree");
    d_grid_ = nullptr;
    check(hipMalloc(&d_grid_, elem_size() * cnt), "hipMalloc(grid)");
    if (dtype_ == TOLFG_F64) {
        blocking_upload(d_grid_, g.v, sizeof(double) * cnt, "upload(grid)");
    } else {
        std::vector<float> tmp(g.v, g.v + cnt);
        blocking_upload(d_grid_, tmp.data(), sizeof(float) * cnt, "upload(grid)");
    }
    grid_host_.assign(g.v, g.v + cnt);      // for the opt-in Woutput.txt dump (problem::dump_wind)
    GridDev &d = args_.grid;
    d.v = d_grid_; d.nx = g.nx; d.ny = g.ny; d.nz = g.nz; d.pad = 0;
    d.x0 = g.x0; d.y0 = g.y0; d.z0 = g.z0; d.dx = g.dx; d.dy = g.dy; d.dz = g.dz;
    d.e0 = g.east_from_datum; d.n0 = g.north_from_datum; d.u0 = g.up_from_datum;
    windmodel_ = TOLFG_WIND_GRID;
}

bool batch::grid_wind_host(double pn, double pe, double pd, double *v, double *dve, double *dvn, double *dvu) const
{
    // the kernel's NodeCtx::grid_wind in host doubles (ref: src/problem.cpp:551-692)
    if (grid_host_.empty()) return false;
    const GridDev &gr = args_.grid;
    const double xs = pe + gr.e0, ys = pn + gr.n0, zs = -pd + gr.u0;
    auto cell = [](double s, double o, double d, int n) { return std::min(std::max((int)std::floor((s - o) / d), 0), n - 2); };
    const int xi = cell(xs, gr.x0, gr.dx, gr.nx), yi = cell(ys, gr.y0, gr.dy, gr.ny), zi = cell(zs, gr.z0, gr.dz, gr.nz);
    const double *g = grid_host_.data() + ((long)xi * gr.ny + yi) * gr.nz + zi;
    const long sx = (long)gr.ny * gr.nz, sy = gr.nz;
    const double c[8] = {g[0], g[sx], g[sy], g[sx + sy], g[1], g[sx + 1], g[sy + 1], g[sx + sy + 1]};
    const double ze = (xs - (gr.x0 + xi * gr.dx)) / gr.dx, et = (ys - (gr.y0 + yi * gr.dy)) / gr.dy, mu = (zs - (gr.z0 + zi * gr.dz)) / gr.dz;
    const double a = 1 - ze, b = 1 - et, m = 1 - mu;
    *v = a * b * m * c[0] + ze * b * m * c[1] + a * et * m * c[2] + ze * et * m * c[3] + a * b * mu * c[4] + ze * b * mu * c[5] + a * et * mu * c[6] + ze * et * mu * c[7];
    *dve = ((c[1] - c[0]) * b * m + (c[3] - c[2]) * et * m + (c[5] - c[4]) * b * mu + (c[7] - c[6]) * et * mu) / gr.dx;
    *dvn = ((c[2] - c[0]) * a * m + (c[3] - c[1]) * ze * m + (c[6] - c[4]) * a * mu + (c[7] - c[5]) * ze * mu) / gr.dy;
    *dvu = ((c[4] - c[0]) * a * b + (c[5] - c[1]) * ze * b + (c[6] - c[2]) * a * et + (c[7] - c[3]) * ze * et) / gr.dz;
    return true;
}

double batch::chi_d(int t) const
{
    const tolfg_traj &tr = host_traj_.at(t);
    return std::atan2(tr.east_goal - tr.yi, tr.north_goal - tr.xi);   // ref: src/problemG7.cpp:524
}

void batch::set_trajectories(int B, const tolfg_traj *trajs)
{
    if (B < 1 || !trajs) throw std::invalid_argument("set_trajectories: B >= 1 and a table are required");
    std::vector<TrajDev> dev(B);
    for (int t = 0; t < B; ++t) {
        const tolfg_traj &tr = trajs[t];
        if (tr.aircraft < 0 || tr.aircraft >= (int)acs_.size()) throw std::invalid_argument("aircraft index");
        if (mission_ == MISSION_MIXED && tr.mission != MISSION_S10 && tr.mission != MISSION_G7)
            throw std::invalid_argument("trajectory mission must be 0 (S10) or 1 (G7) in a mixed batch");
        TrajDev &d = dev[t];
        d.shear = tr.Vref / tr.href;
        d.xg = tr.north_goal; d.yg = tr.east_goal; d.rg = tr.radius_goal;   // ENU -> NED
        const double cd = std::atan2(tr.east_goal - tr.yi, tr.north_goal - tr.xi);
        d.cchi = std::cos(cd); d.schi = std::sin(cd);
        d.ac = tr.aircraft;
        d.mission = mission_ == MISSION_MIXED ? tr.mission : mission_;
        d.xi = tr.xi; d.yi = tr.yi; d.zi = tr.zi; d.chi_d = cd;
    }
    host_traj_.assign(trajs, trajs + B);
    dev_traj_.swap(dev);
    ntraj_ = B;
    uploaded_ = false;     // the device copy is made by the first eval (set-up needs no GPU)
}

void batch::upload(hipStream_t stream)
{
    // Everything the kernels read besides X -- the trajectory table and the initial guess's per-node table -- goes to the
    // device ON THE STREAM OF THE CALL THAT NEEDS IT, from pinned staging, and that stream is drained before this returns:
    // nothing here rides the null stream (which is not ordered against a non-blocking launch stream), and a later call on
    // any other stream finds the uploads complete.  Rare (once per set_trajectories) and blocking; never inside a capture.
    DeviceGuardLocal guard;
    check(hipSetDevice(device_), "hipSetDevice");
    if (cus_ <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess && cus > 0) cus_ = cus;
        else { cus_ = 256; clear_errors(); }
    }
    if (ntraj_ > cap_) {
        if (d_traj_) check(hipFree(d_traj_), "hipFree");
        d_traj_ = nullptr;
        check(hipMalloc(reinterpret_cast<void **>(&d_traj_), sizeof(TrajDev) * (size_t)ntraj_), "hipMalloc(traj)");
        cap_ = ntraj_;
    }
    const size_t traj_bytes = sizeof(TrajDev) * (size_t)ntraj_;
    const int N = sz_.N;
    const size_t ld = (size_t)N + 1;
    const size_t tab_bytes = d_tgrid_ ? 0 : sizeof(double) * 2 * (size_t)X0_FIELDS * ld;
    void *stage = nullptr;
    check(hipHostMalloc(&stage, traj_bytes + tab_bytes, hipHostMallocDefault), "hipHostMalloc(upload staging)");
    try {
        std::memcpy(stage, dev_traj_.data(), traj_bytes);
        check(hipMemcpyAsync(d_traj_, stage, traj_bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(traj)");
        if (tab_bytes) {
            // node times exactly as InitialCond forms them: t = t + dt from 0 (ref: src/problemS10.cpp:60-64); the rest of
            // the table (what a node's row holds for every trajectory of a mission alike) is computed on the device from
            // them, once per batch object
            double *tg = reinterpret_cast<double *>(static_cast<char *>(stage) + traj_bytes);
            std::memset(tg, 0, tab_bytes);
            for (int m = 0; m < 2; ++m) {
                const double dt = (m == 0 ? 20.0 : 10.0) / N;
                double t = 0.0;
                for (int k = 0; k <= N; ++k, t = t + dt) tg[(size_t)m * X0_FIELDS * ld + k] = t;
            }
            check(hipMalloc(reinterpret_cast<void **>(&d_tgrid_), tab_bytes), "hipMalloc(x0 table)");
            check(hipMemcpyAsync(d_tgrid_, tg, tab_bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(x0 table)");
            check(launch_x0_table(d_tgrid_, N, mission_, stream), "launch x0 table");
        }
        if (!h_status_) {
            check(hipHostMalloc(reinterpret_cast<void **>(&h_status_), 64, hipHostMallocMapped), "hipHostMalloc(status)");
            *h_status_ = 0;
            check(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_status_), h_status_, 0), "hipHostGetDevicePointer(status)");
        }
        check(hipStreamSynchronize(stream), "hipStreamSynchronize(upload)");
    } catch (...) {
        (void)hipStreamSynchronize(stream);
        (void)hipHostFree(stage);
        throw;
    }
    check(hipHostFree(stage), "hipHostFree(upload staging)");
    uploaded_ = true;
}

bool batch::take_lost_partial()
{
    if (!h_status_ || __atomic_load_n(h_status_, __ATOMIC_ACQUIRE) == 0) return false;
    __atomic_store_n(h_status_, 0u, __ATOMIC_RELEASE);
    return true;
}

void batch::eval(int B, const void *dX, long ldx, void *dF, long ldf, void *dG, long ldg, const void *dWind,
                 int needF, int needG, hipStream_t stream, void *dObj, unsigned long long *done,
                 unsigned long long done_seq)
{
    if (B < 1 || B > ntraj_) throw std::invalid_argument("eval: B exceeds the described trajectories");
    if (!dX || (needF && !dF) || (needG && !dG)) throw std::invalid_argument("eval: null device pointer");
    if (ldx < sz_.n || (needF && ldf < sz_.neF) || (needG && ldg < sz_.neG))
        throw std::invalid_argument("eval: leading dimension smaller than the row");
    if (windmodel_ == TOLFG_WIND_TABLE && !dWind && !store_shape_) throw std::invalid_argument("eval: table wind needs dWind");
    if (windmodel_ == TOLFG_WIND_GRID && !d_grid_ && !store_shape_) throw std::invalid_argument("eval: grid wind needs tolfg_*_set_wind_grid");
    if (!uploaded_) upload(stream);
    // Stream contract (include/tolfg.h): the per-launch workspace (objective partials, arrival counters) belongs to ONE
    // evaluation at a time.  Evaluations on one stream are ordered by the stream; a caller that moves to another stream
    // gets the ordering enforced here -- the previous stream is drained first (a rare, blocking event, no cost otherwise).
    // (Not while `stream` is being captured into a hipGraph: nothing executes then, and a synchronisation would break the
    // capture; the caller warmed up and synchronised before capturing, as the header asks.)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) { capturing = hipStreamCaptureStatusNone; clear_errors(); }
    if (capturing == hipStreamCaptureStatusNone) {
        if (have_last_stream_ && stream != last_stream_) check(hipStreamSynchronize(last_stream_), "hipStreamSynchronize(previous stream)");
        last_stream_ = stream;
        have_last_stream_ = true;
    }
    if (take_lost_partial())
        throw hip_failure("an earlier evaluation of this batch lost an objective partial: its F[0] is not a number (does x carry "
                          "NaNs?); the outputs of that evaluation must not be used");
    FgArgs a = args_;
    // a handful of short trajectories (the SNOPT callback is B = 1): one launch, whole trajectory per
    // workgroup (measured per call: ts=100 24.7 vs 29.2 us, ts=200 29.7 vs 33.1 us; at ts=500, 8 waves
    // per workgroup, the tile-per-workgroup path is as fast, so the single form is used up to ts = 256)
    const bool no_single = no_single_, force_single = force_single_;      // measurement build only (knobs.h)
    // (the placement probe plans for the launch it stands in for, F + G, whatever it asks of this call: alloc_outputs)
    const double out_bytes = plan_out_bytes_ > 0 ? plan_out_bytes_ : (double)elem_size() * B * ((needF ? sz_.neF : 0) + (needG ? sz_.neG : 0));
    // 16-byte window loads and defect stores need the rows of X and F on 16-byte boundaries; the slab stream
    // copes with any position of G (the waves shift their streams)
    const int vmax = dtype_ == TOLFG_F64 ? 2 : 4;
    const bool aligned = (reinterpret_cast<uintptr_t>(dX) % 16 == 0) && (ldx % vmax == 0) &&
                         (!needF || ((reinterpret_cast<uintptr_t>(dF) % 16 == 0) && (ldf % vmax == 0)));
    const LaunchPlan lp = plan_launch(LaunchShape{B, a.N, dtype_, a.pattern, mission_, aligned ? 1 : 0, out_bytes, needG ? 1 : 0, cus_});
    a.single = (!no_single && (lp.single || (force_single && B <= 8 && a.N <= 256 && mission_ != MISSION_MIXED))) ? 1 : 0;
    int max_nt = a.single ? 0 : (tile_nodes_forced_ > 0 ? tile_nodes_forced_ : lp.max_nt);
    if (!aligned && max_nt > 64) max_nt = 64;          // two nodes per lane (fp32 tiles beyond 64 nodes) need 16-byte rows
    plan_tiles(a.N, dtype_, max_nt, &a.tiles, &a.nt);
    // finer tiles for the trajectories the launch reaches last (FgArgs::tail_count)
    a.tail_count = a.tail_tiles = a.tail_nt = 0;
    const int tail_count = a.single ? 0 : std::min(B, tail_forced_ >= 0 ? tail_forced_ : lp.tail_count);
    if (tail_count > 0) {
        const int tnt = tail_nt_forced_ > 0 ? tail_nt_forced_ : lp.tail_nt;
        a.tail_count = tail_count;
        plan_tiles(a.N, dtype_, std::min(tnt, a.nt), &a.tail_tiles, &a.tail_nt);
    }
    const long body = (long)(B - a.tail_count) * a.tiles;
    const long W = body + (long)a.tail_count * a.tail_tiles;
    if (W > partial_cap_) {        // objective partials, 2 doubles per tile
        check(hipSetDevice(device_), "hipSetDevice");
        if (d_partial_) check(hipFree(d_partial_), "hipFree");
        d_partial_ = nullptr;
        check(hipMalloc(reinterpret_cast<void **>(&d_partial_), sizeof(double) * 2 * (size_t)W), "hipMalloc(partial)");
        // fused path: every slot starts empty and is emptied again by the wave that read it.  ON THE LAUNCH STREAM: a memset on
        // the null stream is not ordered against a non-blocking stream, and a launch that overtook it found its arrival
        // counters zeroed under its feet -- one trajectory in a few thousand first evaluations went without its finalizing
        // wave (objective and boundary rows not written; caught by tests/test_multi_loopback.py, whose parts use such streams)
        check(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_partial_), kEmptySlotWord, 4 * (size_t)W, stream), "hipMemsetD32Async(partial)");
        partial_cap_ = W;
        partial_dirty_ = false;
    }
    if (B > counter_cap_) {        // arrival counters of the fused path (+1: departures): zero between launches
        check(hipSetDevice(device_), "hipSetDevice");
        if (d_counter_) check(hipFree(d_counter_), "hipFree");
        d_counter_ = nullptr;
        check(hipMalloc(reinterpret_cast<void **>(&d_counter_), sizeof(unsigned) * ((size_t)B + 1)), "hipMalloc(counter)");
        check(hipMemsetAsync(d_counter_, 0, sizeof(unsigned) * ((size_t)B + 1), stream), "hipMemsetAsync(counter)");
        counter_cap_ = B;
    }
    a.partial = d_partial_;
    a.counter = d_counter_;
    a.status = d_status_;
    a.fused = ((fused_forced_ >= 0 ? fused_forced_ : lp.fused) || done) ? 1 : 0;   // a completion word needs the single-launch form
    // the two-launch form leaves its partial sums in the slots; the single-launch form polls for slots that are still
    // "empty", so a batch that changes form between evaluations (compact pattern across the cache threshold) refills them
    if (a.fused && partial_dirty_) {
        check(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_partial_), kEmptySlotWord, 4 * (size_t)partial_cap_, stream),
              "hipMemsetD32Async(partial)");
        partial_dirty_ = false;
    }
    if (!a.fused && needF) partial_dirty_ = true;
    a.obj = dObj;
    a.done = done; a.done_seq = done_seq;
    a.waves_per_cu = waves_forced_ ? waves_per_cu_ : lp.waves_per_cu;
    a.nt_stores = nt_forced_ >= 0 ? nt_forced_ : lp.nt_stores;
    a.stagger = (stagger_forced_ >= 0 ? stagger_forced_ : lp.stagger) && !a.single ? 1 : 0;
    const int kwind = kernel_wind(windmodel_);
    a.sub_nodes = (!a.single && dtype_ == TOLFG_F64 && a.pattern == PATTERN_REFERENCE && (kwind == WIND_NONE || kwind == WIND_SHEAR) && a.nt > 32 &&
                   a.nt <= kTileNodes) ? (sub_forced_ >= 0 ? sub_forced_ : lp.sub_nodes) : 0;
    a.xcd_chunk = ((xcd_forced_ >= 0 ? xcd_forced_ : lp.xcd) && !a.single) ? (int)(a.tail_count ? body / 8 : (W + 7) / 8) : 0;
    a.X = dX; a.ldx = ldx; a.F = dF; a.ldf = ldf; a.G = dG; a.ldg = ldg;
    a.wind = dWind; a.traj = d_traj_;
    a.B = B; a.needF = needF ? 1 : 0; a.needG = needG ? 1 : 0;
    a.store_shape = (store_shape_ && !a.single && needG) ? 1 : 0;
    const int vec = aligned ? vmax : 1;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (timing_ && !a.store_shape) {   // HIP events on the launch stream, around the whole evaluation
        if (ev_used_ + 2 > ev_.size()) {
            const size_t old = ev_.size();
            ev_.resize(old + 64, nullptr);
            for (size_t i = old; i < ev_.size(); ++i) check(hipEventCreate(&ev_[i]), "hipEventCreate");
        }
        t0 = ev_[ev_used_++];
        t1 = ev_[ev_used_++];
    }
    check(launch_fg(a, mission_, kernel_wind(windmodel_), dtype_, vec, stream, t0, t1), "launch fg");
}

void batch::set_timing(bool on)
{
    timing_ = on;
    ev_used_ = 0;
}

int batch::kernel_time(double *avg_ms, double *min_ms)
{
    const int pairs = (int)(ev_used_ / 2);
    double sum = 0, mn = 1e300;
    for (int i = 0; i < pairs; ++i) {
        check(hipEventSynchronize(ev_[2 * i + 1]), "hipEventSynchronize");
        float ms = 0;
        check(hipEventElapsedTime(&ms, ev_[2 * i], ev_[2 * i + 1]), "hipEventElapsedTime");
        sum += ms;
        if (ms < mn) mn = ms;
    }
    if (avg_ms) *avg_ms = pairs ? sum / pairs : 0.0;
    if (min_ms) *min_ms = pairs ? mn : 0.0;
    ev_used_ = 0;
    return pairs;
}

void batch::x0_device(int B, void *dX, long ldx, hipStream_t stream)
{
    if (B < 1 || B > ntraj_ || !dX || ldx < sz_.n) throw std::invalid_argument("x0_device: bad arguments");
    if (!uploaded_) upload(stream);          // also builds the per-(mission, node) table, once per batch object
    FgArgs a = args_;
    a.X = dX; a.ldx = ldx; a.traj = d_traj_; a.B = B;
    // x0_serial_ (measurement build, bitwise A/B): the serial reference form, one thread per trajectory
    check(launch_x0(a, mission_, dtype_, x0_serial_ ? nullptr : d_tgrid_, stream), "launch x0");
}

void batch::bounds_device(int B, void *dXlow, void *dXupp, long ldx, void *dFlow, void *dFupp, long ldf, hipStream_t stream)
{
    if (B < 1 || B > ntraj_ || !dXlow || !dXupp || !dFlow || !dFupp || ldx < sz_.n || ldf < sz_.neF)
        throw std::invalid_argument("bounds_device: bad arguments");
    if (!uploaded_) upload(stream);
    BoundsArgs a{};
    a.xlow = dXlow; a.xupp = dXupp; a.ldx = ldx; a.Flow = dFlow; a.Fupp = dFupp; a.ldf = ldf;
    a.traj = d_traj_; a.B = B; a.N = sz_.N; a.mission = mission_;
    for (int ms = 0; ms < 2; ++ms) {
        const limit &lm = limits(ms);
        a.dtmin[ms] = lm.dtmin; a.dtmax[ms] = lm.dtmax;
        for (size_t i = 0; i < acs_.size(); ++i) {      // src/problem.cpp:272-285
            const aircraft &c = acs_[i];
            const double lo[11] = {lm.xmin, lm.ymin, lm.zmin, c.Vamin, -c.gammamax, -1e20, -c.phimax, c.CLmin, -c.phidotmax, -c.phidotmax, c.Tmin};
            const double up[11] = {lm.xmax, lm.ymax, lm.zmax, c.Vamax, c.gammamax, 1e20, c.phimax, c.CLmax, c.phidotmax, c.phidotmax, c.Tmax};
            for (int m = 0; m < 11; ++m) { a.ac[ms][i].lo[m] = lo[m]; a.ac[ms][i].up[m] = up[m]; }
        }
    }
    check(launch_bounds(a, dtype_, stream), "launch bounds");
}

void batch::objectives(int B, const void *dF, long ldf, void *dObj, hipStream_t stream)
{
    if (!dF || !dObj || B < 1) throw std::invalid_argument("objectives: bad arguments");
    check(launch_objectives(dF, ldf, dObj, B, dtype_, stream), "launch objectives");
}

void *batch::alloc_outputs(int B, int tries, long *ldg_out, double *probe_us, int *tried)
{
    if (B < 1 || B > ntraj_) throw std::invalid_argument("alloc_outputs: B exceeds the described trajectories");
    const long v = dtype_ == TOLFG_F64 ? 2 : 4;
    const long ldg = (sz_.neG + v - 1) / v * v;
    const size_t bytes = elem_size() * (size_t)B * (size_t)ldg;
    if (ldg_out) *ldg_out = ldg;
    const double out_bytes = (double)elem_size() * B * ((double)sz_.neF + sz_.neG);
    const LaunchPlan lp = plan_launch(LaunchShape{B, sz_.N, dtype_, args_.pattern, mission_, 1, out_bytes, 1, cus_ > 0 ? cus_ : 256});
    // placement matters to launches that stream beyond the cache (non-temporal form); the others take what they get
    const Knobs &kn = knobs();      // cap on the candidates (16) and early-accept ratio (0.82): fixed in the shipped library
    int n = (tries < 1 || !lp.nt_stores || lp.single) ? 1 : (tries > kn.place_cap ? kn.place_cap : tries);
    if (n > 1) {      // the candidates are held side by side: never more than half of the device's free memory
        size_t free_b = 0, total_b = 0;
        check(hipSetDevice(device_), "hipSetDevice");
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t fit = (free_b / 2) / bytes;
            if ((size_t)n > fit) n = fit < 1 ? 1 : (int)fit;
        } else clear_errors();
    }
    if (probe_us) for (int i = 0; i < (tries > 0 ? tries : 1); ++i) probe_us[i] = 0.0;
    if (tried) *tried = n;
    if (n == 1) return device_alloc(device_, bytes);
    check(hipSetDevice(device_), "hipSetDevice");
    // Rejected candidates stay allocated until a choice is made: freed at once, their physical blocks would come straight back
    // as the next candidate.  The search ends early when a candidate is 18 % faster than the slowest seen: fast and slow class are
    // ~20 % apart (229 vs 285 us for the fp64 headline, 125 vs 157 for its fp32 form) with a middle one in between (259, 140).
    // What the probe borrows from this object -- the store-shape switch, the planning size, the stream contract's "last stream"
    // (the probe's private stream is gone afterwards) -- and what it creates is put back / destroyed on EVERY way out.
    struct Probe {
        batch &b;
        bool was_store_shape, had_prev;
        double was_plan_bytes;
        hipStream_t prev_stream, stream = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<void *> held;
        std::vector<double> held_us;
        explicit Probe(batch &bb) : b(bb), was_store_shape(bb.store_shape_), had_prev(bb.have_last_stream_), was_plan_bytes(bb.plan_out_bytes_),
                                    prev_stream(bb.last_stream_) {}
        void drop(size_t keep) {
            for (size_t i = 0; i < held.size(); ++i)
                if (i != keep) { try { device_free(held[i]); } catch (const std::exception &) {} }
            held.clear();
        }
        ~Probe() {
            if (stream) (void)hipStreamSynchronize(stream);
            b.store_shape_ = was_store_shape;
            b.plan_out_bytes_ = was_plan_bytes;
            b.last_stream_ = prev_stream;
            b.have_last_stream_ = had_prev;
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (stream) (void)hipStreamDestroy(stream);
            drop((size_t)-1);
            clear_errors();
        }
    } pr(*this);
    check(hipStreamCreateWithFlags(&pr.stream, hipStreamNonBlocking), "hipStreamCreate");
    check(hipEventCreate(&pr.e0), "hipEventCreate");
    check(hipEventCreate(&pr.e1), "hipEventCreate");
    store_shape_ = true;
    plan_out_bytes_ = out_bytes;       // the probe launches ask for G alone: they are planned as the F + G launch they stand in for
    double slowest = 0.0, fastest = 0.0;
    for (int i = 0; i < n; ++i) {
        // a candidate that cannot be had (memory) or timed ends the search; the ones already timed stand
        try {
            if (i == kn.place_fail_at) throw hip_failure("device_alloc: injected failure (measurement build)");
            void *cand = device_alloc(device_, bytes);
            pr.held.push_back(cand);
            // the bare store loop reads nothing but the trajectory table: G itself stands in for X
            auto run = [&](int reps) {
                for (int r = 0; r < reps; ++r) eval(B, cand, ldg, nullptr, 0, cand, ldg, nullptr, 0, 1, pr.stream);
            };
            run(3);
            check(hipEventRecord(pr.e0, pr.stream), "hipEventRecord");
            run(10);
            check(hipEventRecord(pr.e1, pr.stream), "hipEventRecord");
            check(hipEventSynchronize(pr.e1), "hipEventSynchronize");
            float ms = 0;
            check(hipEventElapsedTime(&ms, pr.e0, pr.e1), "hipEventElapsedTime");
            const double us = 1e3 * ms / 10;
            pr.held_us.push_back(us);
            if (probe_us) probe_us[i] = us;
            if (us > slowest) slowest = us;
            if (fastest == 0.0 || us < fastest) fastest = us;
            if (tried) *tried = i + 1;
        } catch (const std::exception &) {
            clear_errors();
            if (pr.held_us.empty()) throw;            // not even one candidate: the caller hears why
            if (pr.held.size() > pr.held_us.size()) {   // allocated but not timed: not a candidate
                try { device_free(pr.held.back()); } catch (const std::exception &) {}
                pr.held.pop_back();
            }
            break;
        }
        if (fastest < kn.place_early * slowest) break;
    }
    check(hipStreamSynchronize(pr.stream), "hipStreamSynchronize");
    size_t ibest = 0;
    for (size_t i = 1; i < pr.held_us.size(); ++i)
        if (pr.held_us[i] < pr.held_us[ibest]) ibest = i;
    void *best = pr.held[ibest];
    pr.drop(ibest);
    return best;
}

// ------------------------------------------------------------------------------------ problem

namespace {
std::string root_of(const tolfg_config &cfg) { return cfg.root_path ? std::string(cfg.root_path) : default_root(); }
long even(long v) { return (v + 1) & ~1L; }
batch *make_engine(const tolfg_config &cfg)
{
    if (!cfg.mission || !cfg.aircraft) throw std::invalid_argument("mission and aircraft are required");
    return new batch(cfg.mission, root_of(cfg), {cfg.aircraft}, cfg.ts, cfg.windmodel, TOLFG_F64, cfg.device, cfg.pattern);
}
}  // namespace

problem::problem(const tolfg_config &cfg, int mission_id)
    : eng_(make_engine(cfg)), debug(cfg.debug_dumps != 0), ac(eng_->airframe(0)), gn(eng_->gains()),
      lm(eng_->limits()), sn(eng_->snopt_params())
{
    const Sizes &sz = eng_->sizes();
    if (sz.mission != mission_id) throw std::invalid_argument("mission mismatch");
    n = sz.n; neF = sz.neF; neG = sz.neG;
    east = cfg.east; north = cfg.north; up = cfg.up;
    persistent_arrays_ = cfg.persistent_arrays != 0;
    // goals ENU -> NED (ref: src/problem.cpp:24-27)
    yg = cfg.east_goal; xg = cfg.north_goal; zg = -cfg.up_goal; rg = cfg.radius_goal;
    mission = cfg.mission; aircraft_type = cfg.aircraft;

    tolfg_traj tr{};
    tr.aircraft = 0;
    tr.Vref = cfg.Vref; tr.href = cfg.href;
    tr.north_goal = cfg.north_goal; tr.east_goal = cfg.east_goal; tr.radius_goal = cfg.radius_goal;
    tr.xi = cfg.xi; tr.yi = cfg.yi;
    eng_->set_trajectories(1, &tr);

    iGfun.resize(neG); jGvar.resize(neG);
    make_pattern(sz, iGfun.data(), jGvar.data());
    x.resize(n); xlow.resize(n); xupp.resize(n); Flow.resize(neF); Fupp.resize(neF);
    const Start st{cfg.xi, cfg.yi, cfg.zi};
    initial_guess(sz, ac, st, eng_->chi_d(0), x.data());
    set_limits(sz, ac, lm, st, xlow.data(), xupp.data(), Flow.data(), Fupp.data());

    ldx_ = even(n); ldf_ = even(neF); ldg_ = even(neG);
}

void problem::ensure_device()
{
    // Device state is created by the first evaluation so that set-up (sizes, pattern, x0, bounds)
    // also works on a host without a GPU; evaluation itself has no CPU path.
    if (device_ready_) return;
    {   // measurement build only (knobs.h); the shipped library keeps the defaults of problem.h
        const Knobs &kn = knobs();
        if (kn.callback_staging) zero_copy_ = false;
        if (kn.zero_copy_limit >= 0) zero_copy_limit_ = (size_t)kn.zero_copy_limit;
        if (kn.chunks > 0) nchunks_ = std::min(kChunks, kn.chunks);
        if (kn.no_register) register_user_ = false;
        if (kn.no_flag) use_flag_ = false;
        copy_x_ = kn.callback_copy_x;
    }
    check(hipSetDevice(eng_->device()), "hipSetDevice");
    if (!stream_) check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    if (!hx_) check(hipHostMalloc(reinterpret_cast<void **>(&hx_), sizeof(double) * ldx_, hipHostMallocDefault), "hipHostMalloc");
    if (!hF_) check(hipHostMalloc(reinterpret_cast<void **>(&hF_), sizeof(double) * ldf_, hipHostMallocDefault), "hipHostMalloc");
    if (!hG_) check(hipHostMalloc(reinterpret_cast<void **>(&hG_), sizeof(double) * ldg_, hipHostMallocDefault), "hipHostMalloc");
    if (!done_) {
        check(hipHostMalloc(reinterpret_cast<void **>(&done_), 64, hipHostMallocDefault), "hipHostMalloc");
        *done_ = 0;
    }
    if (!dX_) check(hipMalloc(reinterpret_cast<void **>(&dX_), sizeof(double) * ldx_), "hipMalloc");
    if (!dF_) check(hipMalloc(reinterpret_cast<void **>(&dF_), sizeof(double) * ldf_), "hipMalloc");
    if (!dG_) check(hipMalloc(reinterpret_cast<void **>(&dG_), sizeof(double) * ldg_), "hipMalloc");
    for (int c = 0; c < kChunks; ++c)
        if (!chunk_ev_[c]) check(hipEventCreateWithFlags(&chunk_ev_[c], hipEventDisableTiming), "hipEventCreate");
    device_ready_ = true;
}

problem::~problem()
{
    if (prob == this) prob = nullptr;
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (hx_) (void)hipHostFree(hx_);
    if (hF_) (void)hipHostFree(hF_);
    if (hG_) (void)hipHostFree(hG_);
    if (done_) (void)hipHostFree(done_);
    forget_arrays();
    if (dX_) (void)hipFree(dX_);
    if (dF_) (void)hipFree(dF_);
    if (dG_) (void)hipFree(dG_);
    if (dW_) (void)hipFree(dW_);
    for (int c = 0; c < kChunks; ++c) if (chunk_ev_[c]) (void)hipEventDestroy(chunk_ev_[c]);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void problem::set_wind_table(const double *wind_enu)
{
    if (!wind_enu) throw std::invalid_argument("wind table is null");
    ensure_device();
    const size_t bytes = sizeof(double) * 12 * (size_t)(eng_->sizes().N + 1);
    check(hipSetDevice(eng_->device()), "hipSetDevice");
    if (!dW_) check(hipMalloc(reinterpret_cast<void **>(&dW_), bytes), "hipMalloc(wind)");
    blocking_upload(dW_, wind_enu, bytes, "upload(wind)");
    wind_host_.assign(wind_enu, wind_enu + 12 * (size_t)(eng_->sizes().N + 1));
    eng_->set_windmodel(TOLFG_WIND_TABLE);
    staged_ = false;
}

void problem::set_wind_grid(const tolfg_wind_grid &g)
{
    eng_->set_wind_grid(g);
    staged_ = false;
}

void *problem::device_view(void *p, size_t bytes)
{
    if (!register_user_ || !p || reinterpret_cast<uintptr_t>(p) % 16 != 0) return nullptr;
    for (HostView &v : views_)
        if (v.base == p && v.bytes >= bytes) {
            if (v.dev || v.seen != 1) return v.dev;
            // Second sight of this array in a row under tolfg_config.persistent_arrays: it is one the caller keeps
            // (SNOPT's own x, F and G).  Pin it and map it into the device's address space; a failure is
            // remembered (seen = 2, dev = nullptr).
            v.seen = 2;
            void *dev = nullptr;
            if (hipHostRegister(p, v.bytes, hipHostRegisterMapped) == hipSuccess && hipHostGetDevicePointer(&dev, p, 0) == hipSuccess)
                v.dev = dev;
            clear_errors();
            return v.dev;
        }
    // Without the caller's word that its arrays persist nothing is ever pinned by address: an array that was freed and
    // re-allocated at the same address is indistinguishable from one that was kept, and a stale pinning would have
    // the GPU write into pages the new array no longer owns.
    if (!persistent_arrays_) return nullptr;
    // First sight: remember it, use the staging copy this time (arrays that change from call to call are
    // never registered -- registering costs more than the copy it saves)
    if (views_.size() >= 16) forget_arrays();
    views_.push_back(HostView{p, bytes, nullptr, 1});
    return nullptr;
}

void problem::register_arrays(double *xu, double *Fu, double *Gu)
{
    ensure_device();
    check(hipSetDevice(eng_->device()), "hipSetDevice");
    const struct { double *p; size_t bytes; const char *what; } req[3] = {
        {xu, sizeof(double) * (size_t)n, "x"}, {Fu, sizeof(double) * (size_t)neF, "F"}, {Gu, sizeof(double) * (size_t)neG, "G"}};
    for (const auto &r : req) {
        if (!r.p) continue;
        if (reinterpret_cast<uintptr_t>(r.p) % 16 != 0)
            throw std::invalid_argument(std::string("register_arrays: ") + r.what + " must start on a 16-byte boundary");
        bool known = false;
        for (HostView &v : views_)
            if (v.base == r.p) {
                known = v.dev != nullptr && v.bytes >= r.bytes;
                if (!known && v.dev) { (void)hipHostUnregister(v.base); v.dev = nullptr; }
                if (!known) { v.bytes = r.bytes; v.seen = 1; }
            }
        if (known) continue;
        void *dev = nullptr;
        check(hipHostRegister(r.p, r.bytes, hipHostRegisterMapped), "hipHostRegister");
        const hipError_t e = hipHostGetDevicePointer(&dev, r.p, 0);
        if (e != hipSuccess) { (void)hipHostUnregister(r.p); check(e, "hipHostGetDevicePointer"); }
        bool placed = false;
        for (HostView &v : views_) if (v.base == r.p) { v.dev = dev; v.seen = 2; placed = true; }
        if (!placed) views_.push_back(HostView{r.p, r.bytes, dev, 2});
    }
}

void problem::forget_arrays()
{
    if (stream_) (void)hipStreamSynchronize(stream_);     // nothing in flight may still address them
    for (const HostView &v : views_) if (v.dev) (void)hipHostUnregister(v.base);
    clear_errors();
    views_.clear();
    staged_ = false;
}

int problem::registered_arrays() const
{
    int k = 0;
    for (const HostView &v : views_) k += v.dev != nullptr;
    return k;
}

void problem::stage_and_launch(const double xin[], bool needF, bool needG, double *Fuser, double *Guser, bool caller_keeps_x)
{
    ensure_device();
    check(hipSetDevice(eng_->device()), "hipSetDevice");
    // x: the callback's caller keeps its array untouched until we return, so an array seen twice (SNOPT's own x) is
    // registered like F and G and read by the kernel where it lies; otherwise, and always for the separate
    // modelWind / computeF / computeG entry points (they compare against the copy), x goes through the pinned copy
    const bool direct = zero_copy_ && sizeof(double) * ((size_t)n + neF + neG) <= zero_copy_limit_;
    // (the window loads are 16 bytes wide and may touch element n of a row: only an x of even length is read in place)
    const void *vX = (caller_keeps_x && direct && ldx_ == n && !copy_x_) ? device_view(const_cast<double *>(xin), sizeof(double) * n) : nullptr;
    if (!vX) std::memcpy(hx_, xin, sizeof(double) * n);
    x_copied_ = vX == nullptr;
    void *vF = (needF && Fuser) ? device_view(Fuser, sizeof(double) * neF) : nullptr;
    void *vG = (needG && Guser) ? device_view(Guser, sizeof(double) * neG) : nullptr;
    landF_ = vF ? Fuser : hF_;
    landG_ = vG ? Guser : hG_;
    // Zero-copy: the kernel's stores cross PCIe as posted writes at ~55 GB/s and the completion word
    // follows them, so nothing is copied or synchronised (measured per call, profiles/r02_callback.md:
    // ts=200 22.6 vs 33.6 us staged; ts=2000 52 vs 86 us).  Only problems beyond zero_copy_limit_
    // (64 MB of x+F+G, ts > ~60000) keep device buffers and DMA copies.
    flagged_ = false;
    if (direct) {
        // One trajectory is ~200 KB: launch + PCIe latency dominate, not bandwidth.  The kernel reads
        // x from the pinned copy and writes F, G straight into the caller's (registered) arrays or the
        // pinned staging buffers, then reports through the completion word: no copy commands, no
        // stream synchronisation.
        flagged_ = use_flag_;
        eng_->eval(1, vX ? vX : hx_, ldx_, vF ? vF : hF_, ldf_, vG ? vG : hG_, ldg_, dW_, needF, needG, stream_, nullptr,
                   flagged_ ? done_ : nullptr, flagged_ ? ++seq_ : 0);
    } else {
        check(hipMemcpyAsync(dX_, hx_, sizeof(double) * n, hipMemcpyHostToDevice, stream_), "H2D x");
        eng_->eval(1, dX_, ldx_, dF_, ldf_, dG_, ldg_, dW_, needF, needG, stream_);
        if (needF) check(hipMemcpyAsync(landF_, dF_, sizeof(double) * neF, hipMemcpyDeviceToHost, stream_), "D2H F");
        if (needG && vG) {
            // the DMA engine writes the caller's (registered) array itself: one copy command, nothing to do afterwards
            check(hipMemcpyAsync(Guser, dG_, sizeof(double) * neG, hipMemcpyDeviceToHost, stream_), "D2H G");
        } else if (needG) {
            // G comes back in a few pieces, each followed by an event, so that collect() can copy
            // piece i into the caller's array while piece i+1 is still crossing PCIe
            for (int c = 0; c < nchunks_; ++c) {
                const size_t lo = (size_t)neG * c / nchunks_, hi = (size_t)neG * (c + 1) / nchunks_;
                check(hipMemcpyAsync(hG_ + lo, dG_ + lo, sizeof(double) * (hi - lo), hipMemcpyDeviceToHost, stream_), "D2H G");
                check(hipEventRecord(chunk_ev_[c], stream_), "hipEventRecord");
            }
        }
    }
    chunked_ = !direct && needG && !vG;
    staged_ = true; haveF_ = needF; haveG_ = needG;
}

void problem::wait_done()
{
    if (!flagged_) {
        check(hipStreamSynchronize(stream_), "stream sync");
        return;
    }
    // spin on the completion word; every few thousand polls make sure the stream is still healthy
    const volatile unsigned long long *flag = done_;
    for (unsigned long spins = 1;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq_) break;
        if ((spins & 0x3fff) == 0) {
            const hipError_t q = hipStreamQuery(stream_);
            if (q == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq_) break;
                throw hip_failure("the evaluation finished without reporting completion");
            }
            if (q != hipErrorNotReady) check(q, "stream query");
        }
        __builtin_ia32_pause();
    }
    flagged_ = false;
}

void problem::collect(bool wantF, double F[], bool wantG, double G[])
{
    if (wantG && chunked_) {
        for (int c = 0; c < nchunks_; ++c) {
            const size_t lo = (size_t)neG * c / nchunks_, hi = (size_t)neG * (c + 1) / nchunks_;
            check(hipEventSynchronize(chunk_ev_[c]), "hipEventSynchronize");
            if (c == 0 && wantF && F != landF_) std::memcpy(F, landF_, sizeof(double) * neF);     // F's copy precedes G's on the stream
            std::memcpy(G + lo, hG_ + lo, sizeof(double) * (hi - lo));
        }
        if (eng_->take_lost_partial())
            throw hip_failure("the evaluation lost an objective partial: F[0] is not a number (does x carry NaNs?)");
        return;
    }
    wait_done();
    if (wantF && F != landF_) std::memcpy(F, landF_, sizeof(double) * neF);
    if (wantG && G != landG_) std::memcpy(G, landG_, sizeof(double) * neG);
    if (eng_->take_lost_partial())      // DEFINEGusrfg_ turns this into *Status = -2
        throw hip_failure("the evaluation lost an objective partial: F[0] is not a number (does x carry NaNs?)");
}

void problem::dump(const char *name, const double *v, int len)
{
    // ref: src/DefineFG.cpp:16-21,29-34,41-46 -- same file names and "%.14f" format, opt-in here
    if (FILE *fp = std::fopen(name, "w")) {
        for (int i = 0; i < len; ++i) std::fprintf(fp, "%.14f\n", v[i]);
        std::fclose(fp);
    }
}

void problem::dump_wind(const double *xin)
{
    // ref: the dump at the end of problem::modelWind, src/problem.cpp:740-756 -- one line per node, the twelve
    // ENU member vectors u v w du_dx du_dy du_dz dv_dx dv_dy dv_dz dw_dx dw_dy dw_dz, "%.6f" each
    FILE *fp = std::fopen("Woutput.txt", "w");
    if (!fp) return;
    const int N = eng_->sizes().N;
    const tolfg_traj &tr = eng_->trajectory(0);
    for (int i = 0; i <= N; ++i) {
        double w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const double *s = xin + 11 * i + 1;
        switch (eng_->windmodel()) {
        case TOLFG_WIND_SHEAR: {                       // src/problem.cpp:519-524
            const double zs = -s[2];
            w[1] = -tr.Vref * zs / tr.href;
            w[8] = -tr.Vref / tr.href;
            break;
        }
        case TOLFG_WIND_TABLE:
            for (int f = 0; f < 12 && !wind_host_.empty(); ++f) w[f] = wind_host_[(size_t)f * (N + 1) + i];
            break;
        case TOLFG_WIND_GRID:                          // src/problem.cpp:628-692: v, dv_dx (east), dv_dy (north), dv_dz (up)
            eng_->grid_wind_host(s[0], s[1], s[2], &w[1], &w[6], &w[7], &w[8]);
            break;
        default:
            break;
        }
        for (int f = 0; f < 12; ++f) std::fprintf(fp, f < 11 ? "%.6f " : "%.6f\n", w[f]);
    }
    std::fclose(fp);
}

void problem::evaluate(const double xin[], bool needF, double F[], bool needG, double G[])
{
    if (debug) { dump("Xoutput.txt", xin, n); dump_wind(xin); }
    if (!needF && !needG) return;
    static const bool trace = knobs().trace;     // one line per call on stderr
    if (trace) {
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        stage_and_launch(xin, needF, needG, F, G, true);
        const auto t1 = clk::now();
        if (!chunked_) wait_done();
        else check(hipStreamSynchronize(stream_), "stream sync");
        const auto t2 = clk::now();
        collect(needF, F, needG, G);
        const auto t3 = clk::now();
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "tolfg trace: stage+launch %.1f us, wait %.1f us, copy out %.1f us\n", us(t0, t1), us(t1, t2), us(t2, t3));
        if (debug && needF) dump("Foutput.txt", F, neF);
        if (debug && needG) dump("Goutput.txt", G, neG);
        return;
    }
    stage_and_launch(xin, needF, needG, F, G, true);
    collect(needF, F, needG, G);
    if (debug && needF) dump("Foutput.txt", F, neF);
    if (debug && needG) dump("Goutput.txt", G, neG);
}

void problem::modelWind(const double xin[])
{
    stage_and_launch(xin, true, true);
}

void problem::computeF(const double xin[], double F[])
{
    if (!staged_ || !x_copied_ || !haveF_ || std::memcmp(hx_, xin, sizeof(double) * n) != 0) stage_and_launch(xin, true, false);
    collect(true, F, false, nullptr);
}

void problem::computeG(const double xin[], double G[])
{
    if (!staged_ || !x_copied_ || !haveG_ || std::memcmp(hx_, xin, sizeof(double) * n) != 0) stage_and_launch(xin, false, true);
    collect(false, nullptr, true, G);
}

void problem::writeJSON(const std::string &filename, const double *xs, double final_cost) const
{
    // Keys and nesting as the reference's writer produces them (src/problem.cpp:1293-1357); numbers
    // with 17 significant digits so that a reader recovers every double exactly.
    FILE *fp = std::fopen(filename.c_str(), "w");
    if (!fp) throw std::invalid_argument("cannot open " + filename);
    const int N = eng_->sizes().N;
    auto num = [&](double v) { std::fprintf(fp, "%.17g", v); };
    auto key = [&](const char *k, int indent) { std::fprintf(fp, "%*s\"%s\" : ", indent, "", k); };
    auto field = [&](const char *k, double v, bool last, int indent = 6) { key(k, indent); num(v); std::fprintf(fp, last ? "\n" : ",\n"); };
    std::fprintf(fp, "{\n");
    key("FinalCost", 3); num(final_cost); std::fprintf(fp, ",\n");
    key("aircraft", 3); std::fprintf(fp, "{\n");
    field("AR", ac.AR, false); field("CLmax", ac.CLmax, false); field("CLmin", ac.CLmin, false); field("Cd0", ac.Cd0, false);
    field("S", ac.SS, false); field("Tmax", ac.Tmax, false); field("Tmin", ac.Tmin, false); field("Vamax", ac.Vamax, false);
    field("Vamin", ac.Vamin, false); field("b", ac.b, false); field("dphimax", ac.phidotmax, false); field("e", ac.ee, false);
    field("gammamax", ac.gammamax, false); field("mass", ac.mm, false);
    key("name", 6); std::fprintf(fp, "\"%s\",\n", aircraft_type.c_str());
    field("phimax", ac.phimax, true);
    std::fprintf(fp, "   },\n");
    key("args", 3); std::fprintf(fp, "{\n");
    key("aircraft", 6); std::fprintf(fp, "\"%s\",\n", aircraft_type.c_str());
    field("east", east, false); field("north", north, false);
    key("problem", 6); std::fprintf(fp, "\"%s\",\n", mission.c_str());
    field("rd", rg, false); field("up", up, false); field("xg", xg, false); field("yg", yg, false); field("zg", zg, true);
    std::fprintf(fp, "   },\n");
    key("dt", 3); num(xs[0]); std::fprintf(fp, ",\n");
    key("gains", 3); std::fprintf(fp, "{\n");
    field("kT", gn.kT, false); field("ka", gn.ka, false); field("kdt", gn.kdt, false); field("kp", gn.kp, false); field("kv", gn.kv, true);
    std::fprintf(fp, "   },\n");
    key("limits", 3); std::fprintf(fp, "{\n");
    field("dtmax", lm.dtmax, false); field("dtmin", lm.dtmin, false); field("xmax", lm.xmax, false); field("xmin", lm.xmin, false);
    field("ymax", lm.ymax, false); field("ymin", lm.ymin, false); field("zmax", lm.zmax, false); field("zmin", lm.zmin, true);
    std::fprintf(fp, "   },\n");
    key("problem", 3); std::fprintf(fp, "\"%s\",\n", mission.c_str());
    key("snopt", 3); std::fprintf(fp, "{\n");
    field("feas_tol", sn.feas_tol, false); field("numbounds", sn.numbounds, false); field("numinp", sn.numinp, false);
    field("numstates", sn.numstates, false); field("opt_tol", sn.opt_tol, false); field("ts", N, true);
    std::fprintf(fp, "   },\n");
    key("trajectory", 3); std::fprintf(fp, "{\n");
    // JsonCpp orders keys bytewise: upper case first
    const char *names[12] = {"CL", "T", "Va", "chi", "dCL", "dphi", "gam", "phi", "time", "x", "y", "z"};
    const int var[12] = {8, 11, 4, 6, 10, 9, 5, 7, -1, 1, 2, 3};     // offset in a node, -1 = time
    for (int a = 0; a < 12; ++a) {
        key(names[a], 6); std::fprintf(fp, "[ ");
        double tm = 0.0;
        for (int k = 0; k <= N; ++k) {
            num(var[a] < 0 ? tm : xs[var[a] + 11 * k]);
            if (k < N) std::fprintf(fp, ", ");
            tm = tm + xs[0];                 // accumulated like the reference does (:1290)
        }
        std::fprintf(fp, a < 11 ? " ],\n" : " ]\n");
    }
    std::fprintf(fp, "   }\n}\n");
    std::fclose(fp);
}

problemS10::problemS10(const tolfg_config &cfg) : problem(cfg, MISSION_S10) {}

problemG7::problemG7(const tolfg_config &cfg)
    : problem(cfg, MISSION_G7), chi_d(std::atan2(cfg.east_goal - cfg.yi, cfg.north_goal - cfg.xi))
{
}

}  // namespace tolfg
