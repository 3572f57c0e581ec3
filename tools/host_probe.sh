#!/bin/bash
# Run ON THE GPU BOX: is the default bench step host-bound on this box?  Prints ms_per_step and the host's enqueue time.
for i in 1 2 3 4; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'fps', d['roofline_fps']['launch_ms'])"
done
nproc; grep -m1 "model name" /proc/cpuinfo; uptime
