cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tstats -o train -- python3 bench.py --mode train --steps 12 --warmup 4 > gpurun_out/tstats.log 2>&1
python3 tools/train_breakdown.py gpurun_out/tstats/train_kernel_trace.csv > gpurun_out/train_breakdown_now.txt
rm -rf gpurun_out/tstats
