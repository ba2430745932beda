#!/bin/bash
# true per-kernel durations of a dozen training steps (rocprofv3 kernel trace): tools/trace_steps.sh <shape> <batch> <tag> [ENV=VAL ...]
S=${1:-taobao10}; B=${2:-1024}; TAG=${3:-trace}; shift 3
OUT=/tmp/$TAG; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d $OUT -o run -- python3 $REPO/tools/pmc_steps.py $S $B 24 > $OUT/log 2>&1
cd $REPO
python3 - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/run_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last 20 steps' kernels: find tower launches
idx = [i for i, r in enumerate(rows) if "k_tower" in r[0]]
idx = idx[-20:]
first, last = idx[0], len(rows) - 1
span = rows[last][2] - rows[first][1]
print("$TAG: %d steps, %.2f us/step wall (first tower start -> last kernel end)" % (len(idx), span / len(idx) / 1e3))
import collections
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for i in range(first, last + 1):
    n = rows[i][0].split("(")[0].replace("void ", "").replace("mamdr::", "")
    dur[n] += rows[i][2] - rows[i][1]; cnt[n] += 1
    if i > first: gap[n] += rows[i][1] - rows[i - 1][2]
for n in dur:
    print("   %-28s x%3d  avg %.2f us   gap before it %.2f us" % (n, cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / max(cnt[n], 1) / 1e3))
PY
