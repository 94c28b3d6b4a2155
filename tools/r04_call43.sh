cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PYR_ONLY="LoG 1.4 + NCC" timeout 300 rocprofv3 --kernel-trace -d /tmp/ktr -o ktr -- python tools/pyr_throughput.py 4 > /tmp/ktr.log 2>&1
grep thr /tmp/ktr.log
db=$(find /tmp/ktr -name "*.db" | head -1)
python tools/trace_overlap.py "$db" 0.3 | tee gpurun_out/trace_overlap_lognnc.txt
