# rocprofv3 kernel trace of the C3 training step -> per-kernel breakdown (gpurun_out/train_breakdown_now.txt) + per-call durations of selected kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tstats -o train -- python3 bench.py --mode train --steps 12 --warmup 4 > gpurun_out/tstats.log 2>&1
python3 tools/train_breakdown.py gpurun_out/tstats/train_kernel_trace.csv > gpurun_out/train_breakdown_now.txt
python3 - <<'PY' > gpurun_out/train_calls.txt
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/tstats/train_kernel_trace.csv")))
for pat in ("mha_core_bwd", "mha_core_lds", "mha_core_kernel", "layernorm_bwd", "cout1", "upsample2x_bwd", "gn_partial", "norm_bwd_apply_kernel<false>"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    if d:
        per = len(d) // 16 if len(d) >= 16 else len(d)
        print(pat, "calls/step", per, "last step:", [round(v, 1) for v in d[-per:]])
PY
rm -rf gpurun_out/tstats
