// knobs.h -- EVERY environment variable this library reads, in one table.
//
// The shipped library (tol_amd/lib/libtolfg.so) reads the three variables of the first group, once per process, and
// nothing else: a stray environment cannot change which kernel a drop-in under SNOPT runs.  The variables of the second
// group exist only in the measurement build (tol_amd/lib/libtolfg_measure.so, compiled with -DTOLFG_MEASURE from the
// same sources; tools/, profiles/experiments and the A/B tests load it through TOLFG_LIBRARY or
// tol_amd.capi.measure_lib()).  There they are read again whenever a batch / problem object is created, so that one
// process can build objects under different settings.
//
//   shipped (documented in include/tolfg.h, "Environment")
//     TOLFG_RCCL_LIBRARY=path         the collective library tolfg_multi loads instead of searching for librccl
//     TOLFG_TRACE=1                   DEFINEGusrfg_ prints one timing line per call on stderr
//     TOLFG_MULTI_SHARED_DEVICES=1    test seam: tolfg_multi accepts a device ordinal more than once; honoured ONLY
//                                     together with TOLFG_RCCL_LIBRARY (real RCCL refuses such a list anyway)
//   measurement build only
//     TOLFG_WAVES_PER_CU=0..32        resident tile waves per CU (plan.cpp's cap)
//     TOLFG_TILE_NODES=4..128         nodes per tile
//     TOLFG_FUSED=0|1                 two launches (fg + finalize) | one
//     TOLFG_NT_STORES=0|1             plain | non-temporal slab stream
//     TOLFG_XCD=0|1                   tile order over the XCDs
//     TOLFG_STAGGER=0|1               issue priorities by SIMD slot
//     TOLFG_SUB_NODES=0|32            LDS passes of a tile's rows
//     TOLFG_TAIL=count:nt             finer tiles for the last `count` trajectories
//     TOLFG_NO_SINGLE_LAUNCH=1        never the one-workgroup-per-trajectory kernel
//     TOLFG_FORCE_SINGLE_LAUNCH=1     that kernel wherever it can run
//     TOLFG_X0_SERIAL=1               x0_device through the serial reference kernel
//     TOLFG_PLACE_CAP=1..64           cap on the placement candidates of alloc_outputs (16)
//     TOLFG_PLACE_EARLY=0..1          its early-accept ratio (0.82; 0 = try them all)
//     TOLFG_PLACED_CHUNK_KIB=64..2^20 physical chunk of device_alloc (2048)
//     TOLFG_PLACE_FAIL_AT=i           fault injection: candidate i (0-based) of alloc_outputs fails to allocate
//     TOLFG_PLACE_SETTLE=0|1          0 = device_alloc hands a fresh block out at once (round 4's form: the driver's pending wipe may
//                                     zero what is written to it); 1 = it settles the block first (problem.cpp: settle_block)
//     TOLFG_MULTI_GATHER_PRIORITY=0|1 tolfg_multi's gather streams at the lowest | the highest (default) stream priority
//     TOLFG_MULTI_SLOT_WAIT=host|stream  tolfg_multi, an objective buffer still read by the gather of four steps back: the issuing thread
//                                     waits for it (host, default) | a wait marker goes into the launch stream (stream)
//     TOLFG_MULTI_SOLO_COMMS=1        tolfg_multi: every part its own one-rank communicator, a device may appear more than once: several
//                                     parts on one GPU under the REAL library, for timing the host side of a step (the gathered vectors
//                                     then hold the part's own block only)
//     TOLFG_CALLBACK_STAGING=1        the callback through explicit H2D / D2H copies instead of host-mapped arrays
//     TOLFG_ZERO_COPY_LIMIT=bytes     size of x+F+G up to which the callback addresses host memory directly
//     TOLFG_CHUNKS=1..6               pieces of G's device-to-host copy on the staged path
//     TOLFG_NO_REGISTER=1             never pin the caller's arrays
//     TOLFG_NO_FLAG=1                 synchronise the stream instead of spinning on the completion word
//     TOLFG_CALLBACK_COPY_X=1         always stage x
#ifndef TOLFG_KNOBS_H_
#define TOLFG_KNOBS_H_

#include <cstddef>
#include <string>

namespace tolfg {

struct Knobs {
    // ---- shipped
    std::string rccl_library;
    bool trace = false;
    bool multi_shared_devices = false;
    // ---- measurement build only; the shipped library keeps these defaults ("not forced")
    int  waves_per_cu = -1;          // -1 = the plan's
    int  tile_nodes = -1;            // -1 = the plan's; value as given (the batch clamps it to what the kernels accept)
    int  fused = -1, nt_stores = -1, xcd = -1, stagger = -1, sub_nodes = -1;
    int  tail_count = -1, tail_nt = 0;
    bool no_single_launch = false, force_single_launch = false;
    bool x0_serial = false;
    int  place_cap = 16;
    double place_early = 0.82;
    size_t placed_chunk = 2u << 20;
    int  place_fail_at = -1;
    int  place_settle = 1;
    int  multi_gather_priority = 1;
    bool multi_slot_wait_on_host = true;
    bool multi_solo_comms = false;
    bool callback_staging = false;
    long zero_copy_limit = -1;       // -1 = the library's (64 MB)
    int  chunks = -1;                // -1 = the library's (2)
    bool no_register = false, no_flag = false, callback_copy_x = false;
};

// The process's knobs.  Shipped build: read on first use, never again.  Measurement build: refresh_knobs() reads the
// environment again (called where a batch / problem object is created).
const Knobs &knobs();
void refresh_knobs();
// true in libtolfg_measure.so
bool measurement_build();

}  // namespace tolfg
#endif
