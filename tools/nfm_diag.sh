mkdir -p gpurun_out/r03y; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03y/d$i -o run -- python3 $GRAFT_REPO_ROOT/tools/graph_bench.py 3 ple,nfm inproc > /tmp/d$i.log 2>&1 < /dev/null
  grep nfm /tmp/d$i.log | cut -c1-100
done
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do db=$(find gpurun_out/r03y/d$i -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py stats $db gpurun_out/r03y/diag_$i.csv < /dev/null; rm -rf gpurun_out/r03y/d$i; done
