timeout -k 10 200 python tools/served_callers.py 2>&1 | grep "callers:" | cut -c1-260
