#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh run (gpurun_out/prof_<tag>/) into committed files:

  profiles/<tag>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` kernel summary, verbatim
  profiles/<tag>_summary.md         per-kernel time, launch geometry and HBM traffic of fg_kernel
  profiles/traffic_latest.json      HBM bytes per fg_kernel launch, read back by bench.py ("traffic")

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are collected in separate passes, both are in KiB, and on gfx950 FETCH_SIZE counts a
128-byte request of a wide (16 B/lane) coalesced read as 64 bytes, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane streaming stores.  The kernel's reads and writes are such accesses.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    with open(path, newline="") as fh:
        return list(csv.DictReader(fh))


def counter_averages(path):
    """`parse_rocprof.py --avg <counter_collection.csv>`: every counter averaged over the fg_kernel dispatches (tools/pmc_pass.sh)."""
    import collections
    acc = collections.defaultdict(list)
    for r in rows(path):
        if "fg_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-32s avg %18.1f  n=%d" % (k, sum(v) / len(v), len(v)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--avg":
        return counter_averages(sys.argv[2])
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    ts = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dtype = sys.argv[4] if len(sys.argv) > 4 else "f64"
    mission = sys.argv[5] if len(sys.argv) > 5 else "mixed"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))

    stats = rows(os.path.join(src, "stats", "stats_kernel_stats.csv"))
    trace = [r for r in rows(os.path.join(src, "stats", "stats_kernel_trace.csv")) if "fg_kernel" in r["Kernel_Name"]]
    fg = [r for r in stats if "fg_kernel" in r["Name"]][0]

    def counter(sub, name):
        vals = [float(r["Counter_Value"]) for r in rows(os.path.join(src, sub, sub + "_counter_collection.csv"))
                if "fg_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
        return vals

    fetch = counter("fetch", "FETCH_SIZE")
    write = counter("write", "WRITE_SIZE")
    fetch_kib = sum(fetch) / len(fetch)
    write_kib = sum(write) / len(write)
    read_bytes = 2.0 * fetch_kib * 1024.0        # gfx950 correction for wide coalesced reads
    write_bytes = write_kib * 1024.0
    elem = 8 if dtype == "f64" else 4
    n = 11 * (ts + 1) + 1
    sizes = {"S10": (8 * ts + 1 + 11, 107 * ts + 37), "G7": (8 * ts + 1 + 12, 105 * ts + 48)}
    if mission == "mixed":      # mission = b mod 2: half the rows of each (batch even)
        neF = (sizes["S10"][0] + sizes["G7"][0]) / 2
        neG = (sizes["S10"][1] + sizes["G7"][1]) / 2
    else:
        neF, neG = sizes[mission]
    alg = elem * batch * (n + neF + neG)
    avg_ns = float(fg["AverageNs"])
    t0 = trace[0]
    md = []
    md.append(f"# rocprofv3 summary `{tag}` -- bench.py, {mission}/ts={ts}/{dtype}, batch {batch} per GPU, 1 MI355X\n")
    extra = "" if dtype == "f64" and mission == "mixed" else f" --dtype {dtype} --mission {mission} --batch {batch}"
    md.append("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 50 --warmup 5 "
              f"--no-cpu-baseline --no-configs --no-calibration{extra}` (tools/profile_gpu.sh); counters from two further passes of the same "
              "command with `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`.\n")
    md.append("## Kernel time (`--stats`)\n")
    md.append("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
    for r in stats[:4]:
        md.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                  f"{float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
    md.append("\n## fg_kernel launch\n")
    md.append(f"- grid {t0['Grid_Size_X']} threads = {int(t0['Grid_Size_X'])//64} workgroups of {t0['Workgroup_Size_X']}; "
              f"rocprofv3's trace columns: VGPR_Count {t0['VGPR_Count']}, Accum_VGPR_Count {t0['Accum_VGPR_Count']}, "
              f"SGPR_Count {t0['SGPR_Count']}, LDS_Block_Size {t0['LDS_Block_Size']} B, Scratch_Size {t0['Scratch_Size']} B "
              f"-- these are the profiler's own accounting (its VGPR figure is not the code object's, and it shows the "
              f"STATIC LDS only).  The code object (tools/isa_report.py) has no scratch; the kernel's LDS "
              f"is dynamic: (nt + 1) x 35 elements are used per workgroup and the launch requests 160 KiB / cap so that at most `cap` "
              f"one-wave workgroups share a CU (tol_amd/csrc/plan.cpp)")
    md.append(f"- algorithmic bytes per launch = {elem} B x {batch} x (n {n} + neF {neF} + neG {neG}) = {alg/1e6:.2f} MB "
              f"({alg/(batch*ts):.1f} B per node)")
    md.append(f"- average duration {avg_ns/1e3:.2f} us  ->  **{alg/avg_ns:.1f} GB/s algorithmic = {alg/avg_ns/80:.1f} % of the "
              f"8 TB/s HBM3E peak** (profiled pass; bench.py's un-profiled HIP-event figure is in BENCH/DESIGN)")
    plain = os.path.join(src, "bench_plain.json")
    if os.path.exists(plain):
        try:
            with open(plain) as fh:
                line = json.loads(fh.read().strip().splitlines()[-1])
            r = line["roofline"]
            md.append(f"- the same command WITHOUT the profiler, same box, same `gpurun` call, run just before: {1e3 * r['kernel_ms']:.2f} us per launch "
                      f"(HIP events over the timed region) = {100 * r['frac']:.1f} % of peak; the bare store loop of the launch's shape on its buffers "
                      f"{r.get('box_stream_shape_us') or float('nan'):.1f} us, vendor fill {r.get('box_fill_GBs') or float('nan'):.0f} GB/s; placement probe "
                      f"{line['config'].get('output_placement', {}).get('probe_us')}.  Dispatch profiling lengthens every launch by 4-16 us "
                      f"(profiles/r02_event_cost.md), and each process places its own output buffer (profiles/r04_allocation_classes.md): the "
                      f"profiled process's store loop is the `store_shape_kernel` row above")
        except (OSError, ValueError, KeyError, IndexError):
            pass
    md.append("\n## HBM traffic of fg_kernel (PMC, per launch, averaged over "
              f"{len(fetch)} / {len(write)} dispatches)\n")
    md.append(f"- FETCH_SIZE {fetch_kib:.1f} KiB raw -> x2 (gfx950 wide-read correction) = {read_bytes/1e6:.2f} MB read; "
              f"algorithmic read {elem*batch*n/1e6:.2f} MB")
    md.append(f"- WRITE_SIZE {write_kib:.1f} KiB = {write_bytes/1e6:.2f} MB written; algorithmic write "
              f"{elem*batch*(neF+neG)/1e6:.2f} MB")
    md.append(f"- total {(read_bytes+write_bytes)/1e6:.2f} MB = {(read_bytes+write_bytes)/alg:.3f} x algorithmic")
    with open(os.path.join(dst, tag + "_summary.md"), "w") as fh:
        fh.write("\n".join(md) + "\n")
    # traffic_latest.json is what bench.py replays beside its HEADLINE: written only for the headline's profile
    tjson = "traffic_latest.json" if os.environ.get("TOLFG_SIDE_PROFILE") != "1" else tag + "_traffic.json"
    with open(os.path.join(dst, tjson), "w") as fh:
        json.dump({"tag": tag, "batch": batch, "ts": ts, "dtype": dtype, "mission": mission, "pattern": "reference",
                   "source": f"tools/profile_gpu.sh {tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                   "hbm_bytes_per_launch": read_bytes + write_bytes,
                   "read_bytes": read_bytes, "write_bytes": write_bytes,
                   "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
                   "algorithmic_bytes": alg, "fg_kernel_avg_ns_profiled": avg_ns}, fh, indent=1)
    print("\n".join(md))


if __name__ == "__main__":
    main()
