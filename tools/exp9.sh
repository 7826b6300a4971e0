mkdir -p gpurun_out/exp9
timeout -k 10 200 python tools/chain_check.py --count 8 --index 0 > gpurun_out/exp9/check.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp9/check.txt
timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 16 > gpurun_out/exp9/stats.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp9/stats.txt
