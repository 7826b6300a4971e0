mkdir -p gpurun_out/exp18
timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 12 --lanes 3 > gpurun_out/exp18/stats3.txt 2>&1
sed -n 18,64p gpurun_out/exp18/stats3.txt
