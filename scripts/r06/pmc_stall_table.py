"""Per-kernel stall breakdown from rocprofv3 --pmc passes (scripts/r06/pmc_stall.sh).  Counters of different passes are combined per kernel NAME as sums over the run.
Columns (per kernel, whole run):
  us/launch    GRBM_GUI_ACTIVE cycles per dispatch at the trace's own duration (from the kernel trace when present)
  MfmaUtil     SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
  wait%        SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (share of resident-wave cycles spent waiting for any instruction's operands: vmcnt / lgkmcnt / exp)
  ldswait%     SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  issue%       SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (a wave had an instruction in issue)
  VALU/MFMA, LDS/MFMA, VMEM/MFMA   instruction-count ratios (per wave instruction); SQ_INSTS_VALU counts the MFMAs too: other vector instructions per MFMA = VALU/MFMA - 1
  TA/MFMA      (TA_TA_BUSY_sum / 256 CUs) / (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs)
  tcpstall     TCP_PENDING_STALL_CYCLES_sum / TA_TA_BUSY_sum   (vector-memory requests stalled behind pending ones per busy texture-addresser cycle)
  ldsconf      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(collections.Counter)
calls = collections.Counter()
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and key not in seen:
            seen.add(key)
            calls[k] += 1


def ratio(a, b, pct=False, fmt="%.2f"):
    if not b or a is None:
        return "-"
    return ("%.1f%%" % (100 * a / b)) if pct else (fmt % (a / b))


print("%-62s %6s %9s %8s %6s %8s %6s %9s %8s %9s %8s %8s %7s" % ("kernel", "disp", "kcyc/disp", "MfmaUtil", "wait%", "ldswait%", "issue%", "VALU/MFMA", "LDS/MFMA", "VMEM/MFMA", "TA/MFMA", "tcpstall", "ldsconf"))
order = sorted(tot.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])
for k, c in order[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    act = c["GRBM_GUI_ACTIVE"] / 8.0
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
    wc = c["SQ_WAVE_CYCLES"]
    print("%-62s %6d %9.1f %8s %6s %8s %6s %9s %8s %9s %8s %8s %7s" % (
        k[:62], calls[k], act / max(calls[k], 1) / 1e3, ratio(mf, act, True), ratio(c["SQ_WAIT_INST_ANY"], wc, True), ratio(c["SQ_WAIT_INST_LDS"], wc, True),
        ratio(c["SQ_ACTIVE_INST_ANY"], wc, True), ratio(c["SQ_INSTS_VALU"], c["SQ_INSTS_MFMA"]), ratio(c["SQ_INSTS_LDS"], c["SQ_INSTS_MFMA"]),
        ratio(c["SQ_INSTS_VMEM"] or (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]), c["SQ_INSTS_MFMA"]), ratio(c["TA_TA_BUSY_sum"] / 256.0, mf),
        ratio(c["TCP_PENDING_STALL_CYCLES_sum"], c["TA_TA_BUSY_sum"]), ratio(c["SQ_LDS_BANK_CONFLICT"], c["SQ_LDS_IDX_ACTIVE"], True)))
print("(counters present: %s)" % ", ".join(sorted({n for c in tot.values() for n in c})))
