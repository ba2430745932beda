"""Average of every counter in a rocprofv3 --pmc rocpd database, per kernel:  python tools/pmc_dump.py <db> [filter]"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "mamdr"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, n, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if flt in k:
        agg[k.split("(")[0].replace("void ", "").replace("mamdr::", "")][n].append(float(v))
for k in sorted(agg):
    print(k, {n: round(sum(v) / len(v), 1) for n, v in sorted(agg[k].items())}, "launches", len(next(iter(agg[k].values()))))
