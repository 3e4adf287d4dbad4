cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_parity_report.jsonl
export DVLA_PARITY_REPORT=$PWD/gpurun_out/r04_parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/gA_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gA_pytest.log
grep -v Warning gpurun_out/gA_pytest.log | tail -12 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/gA_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/gA_smoke.log | cut -c1-200
