for m in none sync sleep events; do timeout 120 python scripts/_dbg_windows.py $m 2>&1 | tail -8; done
