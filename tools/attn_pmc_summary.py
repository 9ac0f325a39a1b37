"""Summarise the passes of tools/attn_pmc.sh: one row per (kernel, grid) with the mean of every counter over its
dispatches (the first 10 of each are dropped as warm-up) and the derived shares."""
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
    for k in ("cross_fold_kernel", "cross_score_kernel_w16", "sa_block_kernel_w16", "sa_eval_kernel"):
        if k in name:
            return k + name[name.index(k) + len(k):].split("(")[0]
    return None


dur = collections.defaultdict(list)
for r in csv.DictReader(open(one("stats/**/*kernel_trace.csv"))):
    k = short(r["Kernel_Name"])
    if k:
        dur[(k, int(r.get("Grid_Size_X") or r["Grid_Size"]), int(r.get("Workgroup_Size_X") or r["Workgroup_Size"]))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3"):
    f = one(p + "/**/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            key = (k, int(r.get("Grid_Size_X") or r["Grid_Size"]), int(r.get("Workgroup_Size_X") or r["Workgroup_Size"]))
            ctr[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for v in ctr.values() for n in v})
mean = lambda v: sum(v[10:]) / max(1, len(v[10:])) if len(v) > 10 else sum(v) / max(1, len(v))  # noqa: E731
with open(os.path.join(out, "attn_pmc_summary.csv"), "w") as fh:
    fh.write("kernel,grid_threads,workgroup,workgroups,dispatches,avg_us,min_us,mfma_busy_share,wave_wait_share,"
             "wave_issue_stall_share,wave_active_share,valu_per_mfma," + ",".join(names) + "\n")
    for key in sorted(dur):
        k, grid, wg = key
        d = sorted(dur[key])
        c = {n: mean(v) for n, v in ctr[key].items()}
        us = mean(dur[key]) / 1e3
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        fh.write('"%s",%d,%d,%d,%d,%.2f,%.2f,%.3f,%.3f,%.3f,%.3f,%.2f,' % (
            k, grid, wg, grid // wg, len(d), us, d[0] / 1e3, busy, c.get("SQ_WAIT_ANY", 0) / wc,
            c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_INSTS_MFMA", 0))) + ",".join("%.0f" % c.get(n, 0) for n in names) + "\n")
print(open(os.path.join(out, "attn_pmc_summary.csv")).read())
