"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KB) counts 128-B read requests at
64 B, so wide coalesced reads are doubled; WRITE_SIZE (KB) is exact for 16-B-per-lane stores.
Usage: python tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01b_pmc_hbm.json
"""
import collections, csv, glob, json, sys


def avg_per_kernel(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = avg_per_kernel(sys.argv[1], "FETCH_SIZE")
write = avg_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "mamdr" not in k:
        continue
    short = k.split("(")[0].replace("void ", "").replace("mamdr::", "")
    fkb, n = fetch.get(k, (0.0, 0))
    wkb, _ = write.get(k, (0.0, 0))
    out[short] = {"launches": n, "FETCH_SIZE_KB_raw": fkb, "WRITE_SIZE_KB": wkb,
                  "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
                  "note": "read side = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
