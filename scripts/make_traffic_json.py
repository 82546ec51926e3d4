"""Builds profiles/conv_hbm_traffic.json from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
scripts/fwd_only.py (bf16, B=32): mean HBM bytes per launch of every convolution instantiation.
FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B; MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.
The counters are in KB."""
import csv, glob, json, sys, collections, re, subprocess, datetime, os
root, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
by = {}
for k, v in agg.items():
    m = re.match(r"(?:void )?(conv3_kernel<[^>]+>|conv_mfma_kernel<1,[^>]+>|conv4_kernel|bb64_kernel)", k)      # bf16 kernels only
    if not m or 'FETCH_SIZE' not in v or 'WRITE_SIZE' not in v:
        continue
    f = sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE']); w = sum(v['WRITE_SIZE']) / len(v['WRITE_SIZE'])
    by[m.group(1)] = {"launches_sampled": len(v['FETCH_SIZE']), "fetch_size_kb_raw_per_launch": round(f, 1), "write_size_kb_per_launch": round(w, 1),
                      "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
def _commit():
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("POPNET_COMMIT")
    except Exception:
        return os.environ.get("POPNET_COMMIT")
json.dump({"measured_at": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "commit": os.environ.get("POPNET_COMMIT") or _commit(),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), scripts/fwd_only.py, bf16 B=32, mean per launch of each conv instantiation",
           "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "by_kernel": by}, open(out, "w"), indent=1)
print(json.dumps(by, indent=1))
