mkdir -p gpurun_out/exp10
timeout -k 10 300 python tools/chain_order.py --count 8 --index 0 > gpurun_out/exp10/order.txt 2>&1; echo "rc=$?"; cat gpurun_out/exp10/order.txt
