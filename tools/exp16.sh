timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 12 --lanes 3 2>&1 | tail -16
timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 12 --lanes 2 2>&1 | tail -16
