"""Per-kernel summary of tools/train_pmc.sh's three passes over the eager train step (tools/ab_train.py):
  python tools/train_pmc_summary.py gpurun_out/<tag> > profiles/<tag>_summary.csv
One row per kernel of the step: dispatches, mean duration (the pass's own kernel trace), the shader clock it ran at
(GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles / 8 / duration -- meaningful for kernels of >= ~10 us, the counter
brackets the dispatch a little wider than the timestamps), the MFMA pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES is summed
over the 1024 SIMDs), the shares of wave cycles spent waiting / issue-stalled / issuing, VALU per MFMA instruction and LDS
bank-conflict cycles per LDS-active cycle."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


for p in ("p1", "p2", "p3"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if p == "p1" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
mean = lambda v: sum(v) / max(1, len(v))  # noqa: E731
print("kernel,dispatches,avg_us,clock_ghz,mfma_busy_share,wave_wait_share,wave_issue_stall_share,wave_active_share,"
      "valu_per_mfma,lds_conflict_per_lds_active,valu_insts,mfma_insts,lds_insts,vmem_insts,salu_insts")
rows = []
for k, cs in ctr.items():
    c = {n: mean(v) for n, v in cs.items()}
    us = mean(dur[k]) / 1e3 if dur[k] else 0.0
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    rows.append((us * len(dur[k]), '"%s",%d,%.1f,%.2f,%.3f,%.3f,%.3f,%.3f,%.1f,%.3f,%.0f,%.0f,%.0f,%.0f,%.0f' % (
        k, len(dur[k]), us, cyc / (us * 1e3) if us else 0, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0,
        c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_INSTS_MFMA", 0)),
        c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0)),
        c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_MFMA", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM", 0),
        c.get("SQ_INSTS_SALU", 0))))
for _, line in sorted(rows, reverse=True):
    print(line)
