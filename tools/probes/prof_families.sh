cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats -d /tmp/fam -o fam -- python tools/profile_families.py > /tmp/fam.log 2>&1
db=$(find /tmp/fam -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/r02e_families.md >/dev/null 2>&1
sed -n 1,40p gpurun_out/r02e_families.md
