mkdir -p gpurun_out/exp19
timeout -k 10 120 python tools/server_check.py --size 480x272 --count 1 --frames 7 --time-frames 50 > gpurun_out/exp19/small.txt 2>&1; echo "small rc=$?"; cat gpurun_out/exp19/small.txt
timeout -k 10 200 python tools/server_check.py --count 8 --index 0 > gpurun_out/exp19/share.txt 2>&1; echo "share rc=$?"; cat gpurun_out/exp19/share.txt
