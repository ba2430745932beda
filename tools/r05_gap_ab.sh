#!/bin/bash
# round 5: kernel durations AND the gaps between consecutive kernels of the headline's step loop, from rocprofv3 kernel traces
# of two trees (usage on the GPU box: tools/r05_gap_ab.sh <dir A> <dir B>; each dir holds bench.py + mamdr_amd/)
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  N=$(basename $T); OUT=$GRAFT_REPO_ROOT/gpurun_out/r05y/gap_$N; rm -rf $OUT; mkdir -p $OUT
  EXTRA=""; grep -q -- '"--lanes"' $T/bench.py && EXTRA="--lanes 0"
  (cd $T && rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 bench.py --no-targets --cpu-budget 0 --no-profile $EXTRA --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/err.txt)
  python3 - "$OUT" "$N" <<'P'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur, gap_after = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0][:40]
    dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000: gap_after[k].append(g)
print("==", name)
for k in sorted(dur, key=lambda k: -sum(dur[k]))[:4]:
    d, g = dur[k], gap_after[k]
    print("  %-42s n %6d  avg dur %7.2f us   avg gap after %6.2f us (median %.2f)" % (k, len(d), sum(d) / len(d) / 1e3, sum(g) / max(len(g), 1) / 1e3, sorted(g)[len(g) // 2] / 1e3 if g else 0))
P
done
