timeout -k 10 200 python tools/chain_modes.py 2>&1
timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 12 --lanes 3 2>&1 | head -18
timeout -k 10 200 python tools/chain_check.py --count 8 --index 0 --frames 9 --lanes 3 --time-frames 50 2>&1 | head -3
