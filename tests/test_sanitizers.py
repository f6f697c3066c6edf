"""AddressSanitizer + UBSan over the CPU code: the oracle's C restatement and
the product's host-side set-up (params.cpp, setup.cpp).  GPU sanitizers are not available on the pool."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "orc_san")
    subprocess.run(["gcc", "-std=c11", "-D_GNU_SOURCE", *SAN, "-I", os.path.join(ROOT, "oracle"),
                    os.path.join(HERE, "c", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "tolfg_oracle.c"),
                    "-o", exe, "-lm"], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "rc=0" in res.stdout


HOST_DRIVER = r'''
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "params.h"
#include "setup.h"
#include "kernels.h"
using namespace tolfg;
int main(int argc, char **argv) {
    const std::string root = argv[1];
    int rc = 0;
    for (const char *m : {"S10", "G7"}) {
        const int mid = std::string(m) == "S10" ? MISSION_S10 : MISSION_G7;
        gain g(m, root); limit l(m, root); snopt s(m, root);
        for (const char *a : {"tempest", "skywalker"}) {
            aircraft ac(a, root);
            for (int pat = 0; pat < 2; pat++)
                for (int N : {1, 5, 64, 201}) {
                    Sizes sz = make_sizes(mid, N, pat);
                    std::vector<int> iG(sz.neG), jG(sz.neG);
                    make_pattern(sz, iG.data(), jG.data());
                    for (int e = 1; e < sz.neG; e++)
                        if ((long)iG[e] * sz.n + jG[e] <= (long)iG[e - 1] * sz.n + jG[e - 1]) rc = 1;
                    std::vector<double> x(sz.n), xl(sz.n), xu(sz.n), Fl(sz.neF), Fu(sz.neF);
                    initial_guess(sz, ac, Start{1, 2, -3}, 0.7, x.data());
                    set_limits(sz, ac, l, Start{1, 2, -3}, xl.data(), xu.data(), Fl.data(), Fu.data());
                    for (int cap : {0, 64, 32, 16, 4}) {
                        int tiles, nt; plan_tiles(N, 0, cap, &tiles, &nt);
                        if (tiles * nt < N || (tiles - 1) * nt >= N || nt > (cap ? cap : 64) || nt % 4) rc = 2;
                    }
                }
        }
    }
    {   // launch plan (plan.cpp): outputs that fit the Infinity Cache vs outputs beyond it
        const LaunchPlan small = plan_launch(LaunchShape{512, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 100e6});
        const LaunchPlan mid = plan_launch(LaunchShape{1024, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 190e6});
        const LaunchPlan big = plan_launch(LaunchShape{4096, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 800e6});
        const LaunchPlan big32 = plan_launch(LaunchShape{4096, 200, 1, PATTERN_REFERENCE, MISSION_S10, 1, 800e6});
        const LaunchPlan bigc = plan_launch(LaunchShape{4096, 200, 0, PATTERN_COMPACT, MISSION_S10, 1, 800e6});
        const LaunchPlan mix32 = plan_launch(LaunchShape{8192, 200, 1, PATTERN_REFERENCE, MISSION_MIXED, 1, 800e6});
        const LaunchPlan mix32u = plan_launch(LaunchShape{8192, 200, 1, PATTERN_REFERENCE, MISSION_MIXED, 0, 800e6});
        if (small.nt_stores || small.waves_per_cu || !small.fused || !small.xcd || small.max_nt != 64 || small.stagger) rc = 4;
        if (!big.nt_stores || big.waves_per_cu != 8 || !big.fused || big32.waves_per_cu != 12 || bigc.fused || big.stagger) rc = 5;
        if (!mid.stagger || mid.nt_stores || big32.max_nt != 64 || mix32.max_nt != 128 || mix32.waves_per_cu != 12 || mix32u.max_nt != 64) rc = 6;
        // round 4: a tile's rows through LDS in two passes for fp64 launches in the cache with 11-20 tile waves per CU; the cache switch at 240 MiB
        const LaunchPlan near = plan_launch(LaunchShape{1280, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 236e6});
        const LaunchPlan far = plan_launch(LaunchShape{1536, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 283e6});
        const LaunchPlan mid32 = plan_launch(LaunchShape{1024, 200, 1, PATTERN_REFERENCE, MISSION_S10, 1, 95e6});
        const LaunchPlan ten = plan_launch(LaunchShape{640, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 118e6});
        if (mid.sub_nodes != 32 || small.sub_nodes || ten.sub_nodes || big.sub_nodes || mid32.sub_nodes || near.sub_nodes != 32 || near.nt_stores ||
            !far.nt_stores || far.sub_nodes) rc = 8;
        // round 4: the callback's single trajectory of 100+ nodes with the Jacobian wanted runs as tiles of <= 28 nodes (>= 5 of them)
        const LaunchPlan cb200 = plan_launch(LaunchShape{1, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 184e3, 1});
        const LaunchPlan cb100 = plan_launch(LaunchShape{1, 100, 0, PATTERN_REFERENCE, MISSION_S10, 1, 92e3, 1});
        const LaunchPlan cbF = plan_launch(LaunchShape{1, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 13e3, 0});
        const LaunchPlan cb60 = plan_launch(LaunchShape{1, 60, 0, PATTERN_REFERENCE, MISSION_S10, 1, 55e3, 1});
        const LaunchPlan few = plan_launch(LaunchShape{4, 200, 0, PATTERN_REFERENCE, MISSION_S10, 1, 740e3, 1});
        if (cb200.single || cb200.max_nt != 28 || cb100.single || cb100.max_nt != 20 || !cbF.single || !cb60.single || !few.single || small.single) rc = 9;
        for (int cap : {128, 100}) {
            int tiles, nt; plan_tiles(200, 1, cap, &tiles, &nt);
            if (tiles != 2 || nt != 100) rc = 7;
            plan_tiles(200, 0, cap, &tiles, &nt);
            if (nt > 64) rc = 7;
        }
    }
    try { aircraft bad("no_such_airframe", root); rc = 3; } catch (const std::length_error &) {}
    std::printf("host sanitize run rc=%d\n", rc);
    return rc;
}
'''


def test_host_setup_under_asan_ubsan(tmp_path):
    src = tmp_path / "host_san.cpp"
    src.write_text(HOST_DRIVER)
    exe = str(tmp_path / "host_san")
    csrc = os.path.join(ROOT, "tol_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", *SAN, "-I", csrc, "-isystem", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    str(src), os.path.join(csrc, "plan.cpp"), os.path.join(csrc, "params.cpp"), os.path.join(csrc, "setup.cpp"), "-o", exe],
                   check=True)
    data = os.path.join(ROOT, "tol_amd", "data") + "/"
    res = subprocess.run([exe, data], capture_output=True, text=True, env=ENV, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "rc=0" in res.stdout


KNOBS_DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include "knobs.h"
using namespace tolfg;
int main() {
    int rc = 0;
    const Knobs &k = knobs();
    if (measurement_build()) {
        // the measurement build: what the environment said when the process started ...
        if (k.fused != 0 || k.tile_nodes != 128 || k.tail_count != 7 || k.tail_nt != 16 || !k.no_flag || k.place_settle != 0 || k.multi_slot_wait_on_host) rc = 1;
        if (k.rccl_library != "/nowhere/librccl.so" || !k.multi_shared_devices || !k.trace) rc = 2;
        // ... and again whenever an object is created
        setenv("TOLFG_FUSED", "1", 1); unsetenv("TOLFG_TILE_NODES"); setenv("TOLFG_WAVES_PER_CU", "99", 1); setenv("TOLFG_RCCL_LIBRARY", "/elsewhere", 1);
        refresh_knobs();
        if (k.fused != 1 || k.tile_nodes != -1 || k.waves_per_cu != 32) rc = 3;
        if (k.rccl_library != "/nowhere/librccl.so") rc = 4;           // the collective library is chosen once per process
    } else {
        // the shipped build: three variables, nothing else, never again
        if (k.fused != -1 || k.tile_nodes != -1 || k.tail_count != -1 || k.no_flag || k.place_settle != 1 || !k.multi_slot_wait_on_host) rc = 5;
        if (k.rccl_library != "/nowhere/librccl.so" || !k.multi_shared_devices || !k.trace) rc = 6;
        setenv("TOLFG_TRACE", "0", 1);
        refresh_knobs();
        if (!k.trace) rc = 7;
    }
    std::printf("knobs run rc=%d\\n", rc);
    return rc;
}
'''


def test_knobs_under_asan_ubsan(tmp_path):
    """tol_amd/csrc/knobs.cpp, both builds: the shipped one reads three variables once; the measurement build the whole table, again
    at every refresh -- except the collective library, which is chosen once per process."""
    src = tmp_path / "knobs_san.cpp"
    src.write_text(KNOBS_DRIVER)
    csrc = os.path.join(ROOT, "tol_amd", "csrc")
    env = dict(ENV, TOLFG_FUSED="0", TOLFG_TILE_NODES="128", TOLFG_TAIL="7:16", TOLFG_NO_FLAG="1", TOLFG_PLACE_SETTLE="0", TOLFG_MULTI_SLOT_WAIT="stream",
               TOLFG_RCCL_LIBRARY="/nowhere/librccl.so", TOLFG_MULTI_SHARED_DEVICES="1", TOLFG_TRACE="1")
    for flags in ([], ["-DTOLFG_MEASURE"]):
        exe = str(tmp_path / ("knobs_san" + ("_measure" if flags else "")))
        subprocess.run(["g++", "-std=c++17", *SAN, *flags, "-I", csrc, str(src), os.path.join(csrc, "knobs.cpp"), "-o", exe, "-lpthread"], check=True)
        res = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
        assert res.returncode == 0 and "rc=0" in res.stdout, res.stdout + res.stderr
        assert "test seam active" in res.stderr                   # the seam announces itself, once
        assert res.stderr.count("test seam active") == 1
