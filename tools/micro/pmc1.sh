cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in exp; do
  rm -rf gpurun_out/pmc_$b
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_$b -o p -- tools/micro/bin/w2d_$b 16 128 128 1024 64 2 2 0 0 3 > /dev/null 2>&1
  echo "== $b"
  python3 tools/pmc_summary.py $(find gpurun_out/pmc_$b -name "*counter_collection.csv") 4 w2d
done
