date
timeout 1500 python tools/fuzz_parity.py --cases 1500 --seed 808080 2>&1 | tail -15 | cut -c1-900
date
