"""Copies a round's rocprofv3 summaries from gpurun_out/ into profiles/<round>/ and derives the per-launch figures the bench's
`roofline` objects cite (HBM traffic with the MI355X guide's FETCH_SIZE correction, SQ counters per wave).  Every summary
records the kernel time per launch of the run the counters were collected on and the git commit of the tree, so that bench.py
can tell whether they still describe the kernel it is timing (counters_for in bench.py).
   usage: python tools/collect_profiles.py [round tag, default r03]      (after tools/profile_all.sh <tag> on the GPU box)"""
import csv, json, os, shutil, subprocess, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", TAG)
os.makedirs(P, exist_ok=True)
GIT = subprocess.run(["git", "-C", R, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
DIRTY = bool(subprocess.run(["git", "-C", R, "status", "--porcelain", "--", "rust-raytracer_amd/csrc"], capture_output=True, text=True).stdout.strip())


def kernel_ms(stats_csv, kernel):
    """average duration (ms) and calls of the kernel in a rocprofv3 --stats summary"""
    for r in csv.DictReader(open(stats_csv)):
        if kernel in r["Name"]:
            return float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    return None, 0

def sums(path, kernel):
    tot, disp = defaultdict(float), set()
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return dict(tot), len(disp)

def slim(src, dst, kernels):
    """keep only the rows of our kernels (the raw CSVs hold one row per counter and dispatch of every kernel)"""
    with open(src) as f, open(dst, "w", newline="") as g:
        rd = csv.reader(f); wr = csv.writer(g)
        wr.writerow(next(rd))
        for row in rd:
            if any(k in row[8] for k in kernels):
                wr.writerow(row)

for tag, name, kernel, cmd in (("c4", "c4_mesh", "k_trace_shade<0, 6,", "python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3"),
                               ("c3", "c3_photon", "k_trace_shade<1, 3,", "python3 bench_extra.py photon")):
    shutil.copy(os.path.join(G, "%s/%s_stats/%s_kernel_stats.csv" % (TAG, tag, tag)), os.path.join(P, "%s_kernel_stats.csv" % name))
    shutil.copy(os.path.join(G, "%s/%s_stats.json.log" % (TAG, tag)), os.path.join(P, "%s_under_rocprof.json.log" % name))
    out = {"kernel": kernel, "command": cmd, "git": GIT + ("+uncommitted kernel sources" if DIRTY else "")}
    avg_ms, calls = kernel_ms(os.path.join(P, "%s_kernel_stats.csv" % name), kernel.rstrip(","))
    out["kernel_ms_per_launch"], out["kernel_calls_in_stats_run"] = avg_ms, calls
    if tag == "c3":  # a photon-mapped frame is several launches (two levels + the heavy-query kernel): what the run itself measured
        try:
            txt = open(os.path.join(P, "%s_under_rocprof.json.log" % name)).read()
            out["kernel_ms_per_frame"] = json.loads(txt[txt.index("{"):txt.rindex("}") + 1])["photon_1m"]["ms"]["render_kernels"]
        except Exception as e:
            out["kernel_ms_per_frame"] = None
            print("c3: no render_kernels figure in the run's output:", e)
    for p in ("fetch", "write", "sq", "sq2"):
        src = os.path.join(G, "pmc_%s_%s/%s/%s_counter_collection.csv" % (TAG, tag, p, p))
        slim(src, os.path.join(P, "%s_pmc_%s.csv" % (name, p)), ("k_trace_shade", "k_gather_heavy", "k_photon_shoot", "k_combine"))
        s, n = sums(src, kernel)
        out.update(s); out["launches_" + p] = n
    n = out["launches_fetch"]
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE (KB) tallies 128-byte requests at 64 B -> doubled; WRITE_SIZE (KB) exact
    out["correction"] = "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read; both in KB in the CSV"
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"] / n + out["WRITE_SIZE"] / out["launches_write"]) * 1024.0
    out["hbm_read_bytes_per_launch"] = 2.0 * out["FETCH_SIZE"] / n * 1024.0
    out["hbm_write_bytes_per_launch"] = out["WRITE_SIZE"] / out["launches_write"] * 1024.0
    w = out["SQ_WAVES"]
    out["per_wave"] = {k: out[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY") if k in out}
    out["per_wave"]["note"] = "SQ_*_CYCLES and SQ_WAIT_* count quad-cycles (x4 for clock cycles)"
    out["valu_lane_utilisation"] = out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_ACTIVE_INST_VALU"] * 64.0) if "SQ_THREAD_CYCLES_VALU" in out else None
    out["wait_fraction_of_wave_cycles"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    out["valu_lane_ops_executed_per_launch"] = out["SQ_THREAD_CYCLES_VALU"] / out["launches_sq2"] if "SQ_THREAD_CYCLES_VALU" in out else None
    json.dump(out, open(os.path.join(P, "%s_pmc_traffic.json" % name), "w"), indent=1)
    print(name, json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "per_wave", "valu_lane_utilisation", "wait_fraction_of_wave_cycles", "valu_lane_ops_executed_per_launch")}))
# memory-pipeline passes of the mesh kernel
mem = {}
for p in ("tcp", "tatd", "sqmem", "tcc", "lat"):
    src = os.path.join(G, "pmc_%s_c4_mem/%s/%s_counter_collection.csv" % (TAG, p, p))
    s, n = sums(src, "k_trace_shade<0, 6,")
    mem.update(s); mem["launches_" + p] = n
mem["l1_hit_rate"] = 1.0 - mem["TCP_TCC_READ_REQ_sum"] / mem["TCP_TOTAL_CACHE_ACCESSES_sum"]
mem["l2_hit_rate"] = mem["TCC_HIT_sum"] / mem["TCC_REQ_sum"]
json.dump(mem, open(os.path.join(P, "c4_mesh_pmc_memory_pipeline.json"), "w"), indent=1)
print("mem", {k: mem[k] for k in ("l1_hit_rate", "l2_hit_rate", "TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "GRBM_GUI_ACTIVE")})
if os.path.exists(os.path.join(G, "%s/bench_default.json.log" % TAG)):
    shutil.copy(os.path.join(G, "%s/bench_default.json.log" % TAG), os.path.join(P, "bench_default.json.log"))
