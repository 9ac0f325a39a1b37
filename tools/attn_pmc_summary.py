"""Summarise the passes of tools/attn_pmc.sh: one row per (kernel, grid, batch) with the mean of every counter over its
dispatches (the first 10 of each are dropped as warm-up) and the derived shares.  The workload launches every (kernel,
batch size) REPS times back to back; a persistent kernel has the same grid at every batch size, so the dispatches of a
(kernel, grid) are cut into runs of REPS in launch order and the run index is reported as `batch_run`."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]


def one(pat):
    f = glob.glob(os.path.join(out, pat), recursive=True)
    return f[0] if f else None


def short(name):
    for k in ("cross_stream_kernel", "cross_fold_kernel", "cross_score_kernel_w16", "sa_block_kernel_w16", "sa_eval_kernel"):
        if k in name:
            return k + name[name.index(k) + len(k):].split("(")[0]
    return None


REPS = int(os.environ.get("REPS", "40"))
dur = collections.defaultdict(list)
seen = collections.Counter()
rows = sorted(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    k = short(r["Kernel_Name"])
    if k:
        base = (k, int(r.get("Grid_Size_X") or r["Grid_Size"]), int(r.get("Workgroup_Size_X") or r["Workgroup_Size"]))
        dur[base + (seen[base] // REPS,)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        seen[base] += 1
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3"):
    f = one(p + "/**/*counter_collection.csv")
    if not f:
        continue
    seen = collections.Counter()
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
        k = short(r["Kernel_Name"])
        if k:
            base = (k, int(r.get("Grid_Size_X") or r["Grid_Size"]), int(r.get("Workgroup_Size_X") or r["Workgroup_Size"]))
            name = r["Counter_Name"]
            ctr[base + (seen[(base, name)] // REPS,)][name].append(float(r["Counter_Value"]))
            seen[(base, name)] += 1
names = sorted({n for v in ctr.values() for n in v})
mean = lambda v: sum(v[10:]) / max(1, len(v[10:])) if len(v) > 10 else sum(v) / max(1, len(v))  # noqa: E731
with open(os.path.join(out, "attn_pmc_summary.csv"), "w") as fh:
    fh.write("kernel,grid_threads,workgroup,workgroups,batch_run,dispatches,avg_us,min_us,mfma_busy_share,wave_wait_share,"
             "wave_issue_stall_share,wave_active_share,valu_per_mfma," + ",".join(names) + "\n")
    for key in sorted(dur):
        k, grid, wg, run = key
        d = sorted(dur[key])
        c = {n: mean(v) for n, v in ctr[key].items()}
        us = mean(dur[key]) / 1e3
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        fh.write('"%s",%d,%d,%d,%d,%d,%.2f,%.2f,%.3f,%.3f,%.3f,%.3f,%.2f,' % (
            k, grid, wg, grid // wg, run, len(d), us, d[0] / 1e3, busy, c.get("SQ_WAIT_ANY", 0) / wc,
            c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_INSTS_MFMA", 0))) + ",".join("%.0f" % c.get(n, 0) for n in names) + "\n")
print(open(os.path.join(out, "attn_pmc_summary.csv")).read())
