#!/usr/bin/env python3
"""Print the essentials of a bench.py JSON line (file argument or stdin)."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(txt.strip().splitlines()[-1])
r = d["roofline"]
nan = float("nan")
print("value %.4g %s on %d GPU(s), %s scaling, backend %s; ms/step %.4f; per launch %.4f ms -> %.0f GB/s = %.3f of peak  (instrumented pass: %.4f ms, min %.4f)"
      % (d["value"], d["unit"], d["n_gpus"], d["scaling"], d.get("backend"), d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"],
         r.get("instrumented_kernel_ms", nan), r.get("instrumented_kernel_min_ms", r.get("kernel_min_ms", nan))))
if "box_fill_GBs" in r:
    print("  this box, same process: vendor fill %.0f GB/s (the launch = %.3f of it); bare store loop of the launch's shape %.1f us = %.0f GB/s over its slab bytes; warm-up %s steps"
          % (r["box_fill_GBs"] or nan, r.get("frac_of_box_fill") or nan, r.get("box_stream_shape_us") or nan, r.get("box_stream_shape_GBs") or nan, d.get("warmup_steps_run")))
for c in d.get("configs", []):
    if c["mode"] == "callback":
        print("  config %d callback: %.1f us/call native, arrays in place (staged: %.1f; via ctypes %.1f; F only %.1f) = %.3g node-evals/s  [%s]"
              % (c["config"], c["us_per_call"], c.get("us_per_call_staged", nan), c["us_per_call_via_python_ctypes"], c.get("us_per_call_needF_only", nan), c["node_evals_per_s"], c["workload"]))
    else:
        extra = "" if c.get("n_gpus", 1) == 1 else " (%d per GPU x %d GPUs, gather %.1f us)" % (c["batch_per_gpu"], c["n_gpus"], c["gather_us"])
        print("  config %d B=%d%s %s: step %.1f us, per launch %.1f us (instrumented %.1f, min %.1f) = %.0f GB/s = %.3f of peak (%.3f of this box's fill; its bare store loop %.1f us), %.3g node-evals/s  [%s]"
              % (c["config"], c["batch"], extra, c["dtype"], 1e3 * c["ms_per_step"], c["eval_us"], c.get("eval_us_instrumented", nan),
                 c.get("eval_min_us_instrumented", c.get("eval_min_us", nan)), c["achieved_GBs"], c["frac_of_hbm_peak"],
                 c.get("frac_of_box_fill") or nan, c.get("box_stream_shape_us") or nan, c["node_evals_per_s"], c["workload"]))
if "cpu_baseline" in d:
    b = d["cpu_baseline"]
    print("  cpu baseline (%s, %d core): %.4g %s; -O0 %.4g; fused 1 core %.4g; fused %d cores %.4g"
          % (b["kind"], b["cores"], b["value"], b["unit"], b["value_O0"], b["fused_one_core"]["value"], b["fused_all_cores"]["cores"], b["fused_all_cores"]["value"]))
if "next_compact_pattern" in d:
    c = d["next_compact_pattern"]
    print("  compact pattern: %.4g node-evals/s, evaluation %.4f ms = %.3f of peak" % (c["value"], c["kernel_ms"], c["frac_of_hbm_peak"]))
if "two_batches_two_streams" in d:
    c = d["two_batches_two_streams"]
    print("  two batches on two streams: %.1f us per evaluation, %.4g node-evals/s = %.3f of peak (whole wall time)"
          % (c["us_per_evaluation"], c["node_evals_per_s"], c["frac_of_hbm_peak"]))
for key in ("s10_batch_4096", "headline_shape_larger_batch"):
    if key in d:
        c = d[key]
        print("  %s: evaluation %.1f us = %.3f of peak, whole step %.4g node-evals/s" % (c["workload"], c["eval_us"], c["frac_of_hbm_peak"], c["node_evals_per_s"]))
if "headline_same_x_every_step" in d:
    c = d["headline_same_x_every_step"]
    print("  side: the headline with the same X buffer every step (x from the Infinity Cache): %.1f us per launch = %.3f of peak over the same algorithmic bytes"
          % (1e3 * c["kernel_ms"], c["frac_of_hbm_peak"]))
for key in ("native_multi", "native_multi_grouped", "native_multi_threads"):
    if key in d:
        c = d[key]
        if "error" in c:
            print("  %s: NO RECORD: %s" % (key, c["error"]))
            continue
        h = c.get("objectives_stored_to_host_instead") or {}
        print("  %s (C++ tolfg_multi, %d device(s), issue %s, rccl %s): step %.1f us (%.4g node-evals/s), per launch %.1f us, without the gather %.1f us; one synchronous gather %.1f us; "
              "host issue %.1f us per step; objectives stored to host instead of gathered: step %.1f us"
              % (key, c["n_gpus"], c.get("issue"), (c.get("rccl") or {}).get("version_code"), 1e3 * c["ms_per_step"], c["node_evals_per_s"], c["eval_us"],
                 c["eval_us_without_gather"], c["gather_us"], c["issue_us_per_step"], 1e3 * h.get("ms_per_step", nan)))
        for r in c.get("configs", []):
            hh = r.get("objectives_stored_to_host_instead") or {}
            print("    config %d B=%d %s: step %.1f us, per launch %.1f us (without the gather %.1f), stored to host instead: %.1f us"
                  % (r["config"], r["batch"], r["dtype"], 1e3 * r["ms_per_step"], r["eval_us"], r["eval_us_without_gather"], 1e3 * hh.get("ms_per_step", nan)))
if "ranks" in d:
    print("  ranks seen: %d; %s" % (d.get("world_seen", 0), "; ".join("rank %d dev %s %s shard %s" % (c["rank"], c["device"], c["pci_bus_id"], c["shard"]) for c in d["ranks"])))
