mkdir -p gpurun_out/exp11
timeout -k 10 200 python tools/chain_check.py --count 8 --index 0 --frames 8 > gpurun_out/exp11/check.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp11/check.txt
timeout -k 10 200 python tools/chain_modes.py > gpurun_out/exp11/modes.txt 2>&1; cat gpurun_out/exp11/modes.txt
timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 12 > gpurun_out/exp11/stats.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp11/stats.txt
