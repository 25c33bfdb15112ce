cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/oc
rocprofv3 --kernel-trace -f csv -d gpurun_out/oc -o oc -- python3 tools/iter_bench.py cfg3 graph 12 > gpurun_out/oc/log 2>&1
python3 tools/overlap_check.py $(find gpurun_out/oc -name "oc_kernel_trace.csv" | head -1)
