#!/bin/bash
# the headline against the internal pass size (frames per pass = workspace size): bench.py --batch B
for b in 256 512 1024 2048; do
  timeout -k 10 300 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null > /tmp/bs_$b.json
  python -c "import json,sys; j=json.loads(open('/tmp/bs_$b.json').read().strip().splitlines()[-1]); print('batch', $b, j['value'], j['ms_per_step'])"
done
