cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/rt -f csv -- python $R/tests/gpu_rollout_trace.py run > $R/gpurun_out/g7_run.log 2>&1
tail -3 $R/gpurun_out/g7_run.log
cd $R
python tests/gpu_rollout_trace.py summary gpurun_out/rt gpurun_out/r04_rollout_step_summary_before.txt
rm -rf gpurun_out/rt
