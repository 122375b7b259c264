timeout 300 python tools/dev/hash_raycast_time.py 2>&1 | grep -E "hash map|Error" | tail -3
