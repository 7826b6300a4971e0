mkdir -p gpurun_out/exp14
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/exp14/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/exp14/pytest.txt
