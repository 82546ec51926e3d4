"""Per-kernel sums of rocprofv3 --pmc passes (profiles/r01_final_pmc_ta_vs_mfma.txt).
usage: pmc_kernel_table.py <dir with one sub-directory per pass>
TA per CU = TA_TA_BUSY_sum / 256, MFMA per SIMD = SQ_VALU_MFMA_BUSY_CYCLES / 1024,
MfmaUtil = MFMA per SIMD / (GRBM_GUI_ACTIVE / 8), LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(collections.Counter)
calls = collections.Counter()
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    seen = set()
    # (each --pmc pass is its own run: counters of different passes are combined per kernel NAME, as sums over the run)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and key not in seen:
            seen.add(key)
            calls[k] += 1
ta_all = sum(c["TA_TA_BUSY_sum"] for c in tot.values())
mf_all = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in tot.values())
print("%-58s %8s %8s %10s %8s %9s %9s" % ("kernel", "dispatch", "TA share", "MFMA share", "TA/MFMA", "MfmaUtil", "LDS confl"))
for k, c in sorted(tot.items(), key=lambda kv: -kv[1]["TA_TA_BUSY_sum"])[:12]:
    ta, mf = c["TA_TA_BUSY_sum"] / 256.0, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
    act = c["GRBM_GUI_ACTIVE"] / 8.0
    print("%-58s %8d %7.1f%% %9.1f%% %8s %8s %9s" % (
        k[:58], calls[k], 100 * c["TA_TA_BUSY_sum"] / max(ta_all, 1), 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(mf_all, 1),
        "%.2f" % (ta / mf) if mf else "-", "%.1f%%" % (100 * mf / act) if mf and act else "-",
        "%.0f%%" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c["SQ_LDS_IDX_ACTIVE"] else "-"))
print("totals: TA_TA_BUSY_sum %.4g  SQ_VALU_MFMA_BUSY_CYCLES %.4g  -> whole run TA per CU / MFMA per SIMD = %.2f" % (ta_all, mf_all, (ta_all / 256) / (mf_all / 1024)))
