timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 10 2>&1 | tail -14
