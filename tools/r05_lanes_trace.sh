#!/bin/bash
# round 5: do the lanes' kernels overlap on the device?  rocprofv3 kernel trace of `bench.py --lanes 4` (single-chain leg, then the
# 4-lane leg in the same process); per hardware queue: kernels and busy time; over the lanes' window: sum of kernel durations /
# wall time = the average number of kernels in flight.   usage (GPU box): tools/r05_lanes_trace.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05y/lanes_trace; rm -rf $OUT; mkdir -p $OUT
(cd $GRAFT_REPO_ROOT && GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 bench.py --no-targets --cpu-budget 0 --no-profile --lanes 4 --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/err.txt)
python3 - "$OUT" <<'P'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
qcol = "Queue_Id" if "Queue_Id" in rows[0] else [c for c in rows[0] if "ueue" in c][0]
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r[qcol]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:36]))
print("hardware queues seen: %d; kernels per queue: %s" % (len(by_q), {q: len(v) for q, v in by_q.items()}))
# the lanes' window: from the first moment two DIFFERENT queues have a tower kernel running to the last such kernel
tower = [(s, e, q) for q, v in by_q.items() for (s, e, n) in v if "k_tower" in n]
qs = sorted(set(q for _, _, q in tower))
main_q = max(qs, key=lambda q: sum(1 for t in tower if t[2] == q))
lane_towers = [t for t in tower if t[2] != main_q]
if lane_towers:
    w0 = min(t[0] for t in lane_towers); w1 = max(t[1] for t in tower if t[0] >= w0)
    inwin = [(s, e, n, q) for q, v in by_q.items() for (s, e, n) in v if s >= w0 and e <= w1]
    busy = sum(e - s for s, e, _, _ in inwin)
    print("lanes window: %.1f ms, %d kernels on %d queues, sum of kernel durations %.1f ms -> %.2f kernels in flight on average" % (
        (w1 - w0) / 1e6, len(inwin), len(set(q for *_, q in inwin)), busy / 1e6, busy / (w1 - w0)))
    d = collections.defaultdict(list)
    for s, e, n, q in inwin: d[n].append(e - s)
    for n in sorted(d, key=lambda n: -sum(d[n]))[:5]:
        print("   %-38s n %6d  avg %7.2f us" % (n, len(d[n]), sum(d[n]) / len(d[n]) / 1e3))
    pre = [(s, e, n) for (s, e, n) in by_q[main_q] if e < w0]
    d2 = collections.defaultdict(list)
    for s, e, n in pre: d2[n].append(e - s)
    print("single-chain leg before it (one queue):")
    for n in sorted(d2, key=lambda n: -sum(d2[n]))[:3]:
        print("   %-38s n %6d  avg %7.2f us" % (n, len(d2[n]), sum(d2[n]) / len(d2[n]) / 1e3))
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("bench under the profiler: single chain %.0f, lanes %.0f domain-steps/s" % (b["value"], b["lanes"]["value"]))
P
