mkdir -p gpurun_out/exp7
timeout -k 10 200 python tools/share_all.py --workload dragon --indices 0 > gpurun_out/exp7/share_all.txt 2>&1; tail -4 gpurun_out/exp7/share_all.txt
timeout -k 10 200 python tools/chain_check.py --count 8 --index 0 --no-check > gpurun_out/exp7/share.txt 2>&1; echo "share rc=$?"; cat gpurun_out/exp7/share.txt
