// multi.h -- one process, several MI355X: a batch of independent trajectories sharded over the devices of one node,
// one launch stream, one gather stream and one issuing host thread per device, and the ONE collective the path has --
// the all-gather of the per-trajectory objectives -- through RCCL over xGMI (BASELINE north star).
// No reference counterpart: tol solves one trajectory per process.
#ifndef TOLFG_MULTI_H_
#define TOLFG_MULTI_H_

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "problem.h"

namespace tolfg {

// Contiguous shard [lo, hi) of `total` trajectories for rank `rank` of `world`: the first total % world ranks get one
// more (the rule of tol_amd/distributed.py::shard_bounds, so both host paths split a batch the same way).
void shard_bounds(long total, int rank, int world, long *lo, long *hi);
// widest shard = shard of rank 0
long shard_width(long total, int world);
// `padded` holds world blocks of `width` values, block r carrying rank r's shard in its first hi_r - lo_r places (what an
// all-gather of equally sized buffers delivers); writes the `total` values in global trajectory order.
void compact_gathered(const void *padded, size_t elem, long total, int world, void *out);

// The RCCL entry points this library uses, resolved at run time (dlopen) so that libtolfg.so has no DT_NEEDED for
// librccl, like it has none for the HIP runtime: a process must hold ONE copy of each, and which copy (PyTorch's
// bundled one, /opt/rocm's) is the host program's choice.
struct rccl_api {
    void *handle = nullptr;
    std::string path;
    int (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int datatype, void *comm, hipStream_t stream) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int datatype, int op, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    static const rccl_api &get();          // throws hip_failure when no usable librccl is found
};

class multi {
public:
    // Objective buffers in rotation (dObj / dAll per device, a pinned host copy on device 0): the evaluation that follows a
    // gather writes the next pair, so a launch never waits for the gather before it -- only for the one kSlots gathers back,
    // which is normally long done (then nothing at all is put into the launch stream).
    static constexpr int kSlots = 4;

    // devices: HIP device ordinals, all different (knobs.h: the shared-devices test seam lifts that).
    // Communicators are created here (ncclCommInitAll).
    multi(const std::string &mission, const std::string &root, const std::vector<std::string> &aircraft_names, int ts,
          int windmodel, int dtype, int pattern, const std::vector<int> &devices);
    ~multi();
    multi(const multi &) = delete;
    multi &operator=(const multi &) = delete;

    int devices() const { return (int)dev_.size(); }
    const Sizes &sizes() const { return part_[0].b->sizes(); }
    long total() const { return total_; }
    void shard(int i, long *lo, long *hi) const { *lo = part_.at(i).lo; *hi = part_.at(i).hi; }

    // describe all `total` trajectories (global order); device i keeps [lo_i, hi_i); (re)allocates X, F, G there.
    // place_tries: candidates of the per-device placement search for G (batch::alloc_outputs); 0 / 1 = none
    void set_trajectories(long total, const tolfg_traj *trajs, int place_tries = 12);
    // device buffers of shard i: rows of the batch dtype in the SNOPT layout, strides in elements
    void buffers(int i, void **dX, long *ldx, void **dF, long *ldf, void **dG, long *ldg) const;
    // wind for every device: one gridded field (wind model 3), or per-trajectory tables [total][12][ts+1] of doubles in global
    // order (TOLFG_WIND_TABLE batches; converted to the batch dtype, device i receives its shard's rows)
    void set_wind_grid(const tolfg_wind_grid &g);
    void set_wind_tables(const double *wind_enu);
    // initial guesses of every shard, generated on its device
    void x0();
    // one evaluation of every shard: one launch per device, issued concurrently by the per-device host threads.
    // dX (optional): devices() device pointers, dX[i] on device i with the row stride buffers() reports -- the shard's
    // x rows are read from there instead of from the library's own X (warm starts; rotating inputs)
    void eval(bool needF, bool needG, const void *const *dX = nullptr);
    // The all-gather of the objectives of the evaluation issued last, enqueued on the devices' gather streams behind an
    // event on their launch streams: returns at once with a ticket; evaluations issued afterwards run beside it.
    unsigned long gather_begin();
    // waits for that gather on every device; out (optional): `total` values of the batch dtype in global trajectory
    // order (device 0's copy).  A ticket stays valid until kSlots further gathers have begun.
    void gather_wait(unsigned long ticket, void *host_out);
    // eval + gather_begin in one go (one wake-up of the issuing threads per step)
    unsigned long step(bool needF, bool needG, const void *const *dX = nullptr);
    // gather_begin + gather_wait + wait for everything (the synchronous form)
    void gather_objectives(void *host_out);
    // Monte-Carlo mean of the objectives: all-reduce(sum) of each device's partial sum (double)
    double mean_objective();
    // device pointer (device i) of the last gather's result: the padded vector, devices() blocks of shard_width() values
    // (GATHER_HOST: the pinned host vector, `total` values in global order, the same address on every device)
    const void *gathered(int i) const;
    void sync();
    std::string rccl_path() const { return rccl_api::get().path; }
    int rccl_version() const;

    // How the collective is issued.  GROUPED: one ncclGroupStart / ncclGroupEnd bracket around the devices' calls, from the
    // caller's thread (the single-process form of the NCCL manual; the default).  THREADS: every device's own issuing thread
    // makes its call for its communicator, no group -- the one-thread-per-device form; with it a step (launch + gather) is
    // issued by each thread on its own, without a rendezvous of the host threads.
    enum { ISSUE_GROUPED = 0, ISSUE_THREADS = 1 };
    void set_issue(int mode);
    int issue() const { return issue_; }

    // Where the objectives are gathered.  RCCL (default): ncclAllGather into a device vector on EVERY device (north star: "RCCL over
    // xGMI only for the final objective gather").  HOST: no collective at all -- every device's finalizing waves store their
    // trajectories' objectives straight into ONE pinned, device-mapped host vector, each shard at its global offset; a gather is
    // then nothing but an event behind the launch (gather_begin) and a wait for it (gather_wait).  For consumers on the host
    // (Monte-Carlo statistics, an SQP driver per trajectory) that takes the collective's launch, its stream hand-over and its
    // 10 us out of every step; the devices do not get each other's objectives.  Switch while nothing is in flight.
    enum { GATHER_RCCL = 0, GATHER_HOST = 1 };
    void set_gather(int mode);
    int gather() const { return gather_; }

    // Measurement aid (bench.py --native-multi): `warm` untimed steps, then `steps` steps -- launch + asynchronous gather,
    // inputs rotating over n_x sets of caller-supplied X buffers ([n_x][devices()] device pointers; n_x = 0: the library's
    // own X) -- between two full synchronisations, issued from native code.
    struct Timing {
        double wall_us_per_step = 0, launch_us_per_step = 0, issue_us_per_step = 0, gather_us = 0;
    };
    Timing time_steps(int n_x, const void *const *dX, bool needF, bool needG, bool gather, int warm, int steps, double *launch_us_per_device);

private:
    struct Part {
        int device = 0, index = 0;
        long lo = 0, hi = 0;
        std::unique_ptr<batch> b;
        hipStream_t stream = nullptr, gstream = nullptr;
        void *dX = nullptr, *dF = nullptr, *dG = nullptr, *dSum = nullptr, *dWind = nullptr;
        void *dObj[kSlots] = {}, *dAll[kSlots] = {};
        hipEvent_t ev_launch[kSlots] = {}, ev_gather[kSlots] = {};
        hipEvent_t t0 = nullptr, t1 = nullptr;
        void *comm = nullptr;
    };
    std::vector<int> dev_;
    std::vector<Part> part_;
    int dtype_;
    int issue_ = ISSUE_GROUPED;
    int gather_ = GATHER_RCCL;
    long total_ = 0, width_ = 0, ldx_ = 0, ldf_ = 0, ldg_ = 0;
    unsigned long seq_ = 0;             // gathers begun so far; the next evaluation's objectives go to slot seq_ % kSlots
    int last_gather_slot_ = 0;
    bool evaluated_since_gather_ = false;
    void *hAll_[kSlots] = {};           // pinned host copies of the gathered, padded objectives
    void *hObj_[kSlots] = {};           // GATHER_HOST: pinned, mapped host vectors [total] the finalizing waves write (global order)
    size_t elem() const { return dtype_ == TOLFG_F64 ? 8 : 4; }
    int nccl_type() const;
    void free_buffers();
    void release();

    // the pieces of a step, each on its part's device (the caller has made it current)
    void part_eval(Part &p, int slot, bool needF, bool needG, const void *X);
    void part_gather_pre(Part &p, int slot);
    void part_gather_call(Part &p, int slot);
    void part_gather_post(Part &p, int slot);
    void steps_run(int n, int n_x, const void *const *dX, bool needF, bool needG, bool gather, unsigned long first_step);

    // one issuing thread per device beyond the first (the caller's thread serves device 0): launches reach the
    // devices side by side instead of one hipSetDevice + launch after the other (8 devices: ~40 us serial)
    void on_every_device(const std::function<void(Part &)> &fn);
    void worker(int i);
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    unsigned long generation_ = 0;
    int pending_ = 0;
    std::atomic<unsigned long> gen_hint_{0};     // copies of generation_ / pending_ for the short spin before a sleep (multi.cpp: spin_until);
    std::atomic<int> pending_hint_{0};           // the mutex-guarded values stay the truth
    bool quit_ = false;
    const std::function<void(Part &)> *job_ = nullptr;
    std::vector<std::string> errors_;
};

}  // namespace tolfg
#endif
