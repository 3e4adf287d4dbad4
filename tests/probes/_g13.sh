timeout 1000 python -m pytest tests/test_model_gpu.py tests/test_shipped_precision_gpu.py tests/test_torch_ddp_gpu.py tests/test_collate.py tests/test_input_pipeline.py tests/test_clip_pin.py -m gpu -q > gpurun_out/r05m_tests.txt 2>&1
tail -4 gpurun_out/r05m_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
