mkdir -p gpurun_out/exp15
timeout -k 10 200 python tools/chain_check.py --count 8 --index 0 --frames 9 --lanes 3 --time-frames 100 > gpurun_out/exp15/check3.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp15/check3.txt
timeout -k 10 200 python tools/chain_check.py --count 8 --index 0 --frames 6 --lanes 2 --time-frames 100 > gpurun_out/exp15/check2.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp15/check2.txt
timeout -k 10 200 python tools/chain_modes.py > gpurun_out/exp15/modes.txt 2>&1; cat gpurun_out/exp15/modes.txt
timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 12 --lanes 3 > gpurun_out/exp15/stats.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp15/stats.txt
