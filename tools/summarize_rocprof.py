"""
Summarise rocprofv3 output (rocpd SQLite, the format this ROCm 7.2 image writes) of tools/probes/prof_run.sh into
profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table) and profiles/<tag>_pmc.json (per-dispatch PMC
averages of the step kernel, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950).

  python tools/summarize_rocprof.py gpurun_out/prof_r1 r1
"""
import csv
import glob
import json
import os
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)

trace = glob.glob(os.path.join(src, "trace", "*.db"))[0]
db = sqlite3.connect(trace)
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2] * 1.0), round(r[3], 1), round(r[4], 4)])
# the SINGLE-STEP kernel of the timed region = the step kernel with the largest total duration among those launched at least as
# often as there are timed steps; the fused-rollout kernel (several control steps per launch, the extra leg of bench.py) and the
# replay kernel (launched behind every step, empty most of the time: lm_step.h) are summarised separately
rows_k = db.execute("select name, count(*) c, sum(duration) d from kernels where name like '%step_kernel%' group by name").fetchall()
cmax = max(r[1] for r in rows_k)
def is_replay(name):          # step_kernel<MC, NS, ...>: the replay kernels are the instantiations with NS > 8 (128 slots per chain)
    try:
        return int(name.split("step_kernel<")[1].split(",")[1]) > 8
    except Exception:
        return False
names = [r[0] for r in sorted(rows_k, key=lambda r: (is_replay(r[0]), r[1] * 2 < cmax, -r[2]))]
kname = names[0]
def kernel_row(name):
    return db.execute("select vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x, min(duration), "
                      "avg(duration), max(duration), count(*) from kernels where name = ?", (name,)).fetchone()
def short(name):
    return "step_kernel" + name.split("step_kernel")[1].split("(")[0]
k = kernel_row(kname)
# registers / spills: the COMPILER's analysis of the built kernel (tools/kernel_resources.py -> profiles/<round>_kernel_resources.json);
# rocprofv3's vgpr_count / accum_vgpr_count columns do not describe these kernels (248 / 0 for one the compiler allocates 256 + 237 for)
# and are kept only under their own names
summary = dict(kernel=short(kname), rocprof_vgpr_count=k[0], rocprof_accum_vgpr_count=k[1], rocprof_sgpr_count=k[2], lds_bytes=k[3], scratch_bytes=k[4],
               workgroup=k[5], grid=k[6], duration_ns=dict(min=k[7], avg=k[8], max=k[9]), dispatches=k[10])
_res = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag.split("_")[0] + "_kernel_resources.json")
if os.path.exists(_res):
    _r = json.load(open(_res))["kernels"].get(short(kname))
    if _r:
        summary["resources"] = dict(_r, source="profiles/%s (hipcc -Rpass-analysis=kernel-resource-usage)" % os.path.basename(_res))
for other in names[1:]:
    o = kernel_row(other)
    summary.setdefault("other_step_kernels", []).append(dict(kernel=short(other), dispatches=o[10], duration_ns=dict(min=o[7], avg=o[8], max=o[9])))
pmc = {}
for p in sorted(glob.glob(os.path.join(src, "pmc*", "*.db"))):
    d = sqlite3.connect(p)
    for name, total, n in d.execute("select counter_name, sum(value), count(*) from counters_collection "
                                    "where kernel_name = ? group by counter_name", (kname,)):
        pmc[name] = dict(per_dispatch=total / n, dispatches=n)
if "FETCH_SIZE" in pmc:
    pmc["FETCH_SIZE"]["unit"] = "KiB"
    pmc["FETCH_SIZE"]["bytes_per_dispatch_corrected_x2"] = pmc["FETCH_SIZE"]["per_dispatch"] * 1024 * 2
if "WRITE_SIZE" in pmc:
    pmc["WRITE_SIZE"]["unit"] = "KiB"
    pmc["WRITE_SIZE"]["bytes_per_dispatch"] = pmc["WRITE_SIZE"]["per_dispatch"] * 1024
summary["pmc"] = pmc
bench = os.path.join(src, "bench_trace.json")
if os.path.exists(bench):
    try:
        summary["bench_line_under_trace"] = json.loads(open(bench).read().strip().splitlines()[-1])
    except Exception:
        pass
# the build these counters belong to: bench.py quotes them only when the library it loads has the same hash
import hashlib
lib = os.path.join(root, "loco_mujoco_amd", "csrc", "liblocohip.so")
if os.path.exists(lib):
    summary["lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
json.dump(summary, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
