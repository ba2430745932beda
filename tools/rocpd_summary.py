"""Summaries from rocprofv3's rocpd (sqlite) output, for runs made without --output-format csv.

  python tools/rocpd_summary.py stats <run_results.db> <out.csv>
        per-kernel Calls / TotalDurationNs / AverageNs / Percentage / MinNs / MaxNs
        (the same columns as rocprofv3 --stats' kernel_stats.csv)
  python tools/rocpd_summary.py pmc <fetch.db> <write.db> <out.json>
        per-kernel HBM bytes per launch from separate FETCH_SIZE / WRITE_SIZE passes with the
        gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B)
  python tools/rocpd_summary.py pmc1 <fetch.db> <write.db> <kernel substring> <key> <out.json>
        the same for ONE kernel, merged into an existing summary under <key>
"""
import collections, csv, json, sqlite3, sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    return [(r[0], r[1], r[2], r[3], 100.0 * r[2] / total, r[4], r[5]) for r in rows]


def counter_avg(db, counter):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(list)
    for name, value in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        agg[name].append(float(value))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def short(k):
    return k.split("(")[0].replace("void ", "").replace("mamdr::", "")


def main():
    if sys.argv[1] == "stats":
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in kernel_stats(sys.argv[2]):
                w.writerow(list(r))
        return
    fetch, write = counter_avg(sys.argv[2], "FETCH_SIZE"), counter_avg(sys.argv[3], "WRITE_SIZE")
    if sys.argv[1] == "pmc1":
        sub, key, path = sys.argv[4], sys.argv[5], sys.argv[6]
        try:
            out = json.load(open(path))
        except Exception:
            out = {}
        for k in sorted(set(fetch) | set(write)):
            if sub in k:
                fkb, n = fetch.get(k, (0.0, 0))
                wkb, _ = write.get(k, (0.0, 0))
                out[key] = {"launches": n, "FETCH_SIZE_KB_raw": fkb, "WRITE_SIZE_KB": wkb,
                            "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
                            "note": "read side = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B)"}
        json.dump(out, open(path, "w"), indent=1)
        return
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "mamdr" not in k:
            continue
        fkb, n = fetch.get(k, (0.0, 0))
        wkb, _ = write.get(k, (0.0, 0))
        out[short(k)] = {"launches": n, "FETCH_SIZE_KB_raw": fkb, "WRITE_SIZE_KB": wkb,
                         "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
                         "note": "read side = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B)"}
    json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
