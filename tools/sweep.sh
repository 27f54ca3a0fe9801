#!/bin/bash
# usage: tools/sweep.sh  -- sweeps LIDARREG_NN_BLOCKS x pairs in flight (development tool)
for b in 128 256 512 1024; do
  for s in 1 16; do
    v=$(LIDARREG_NN_BLOCKS=$b python bench.py --no-cpu-baseline --streams $s --pairs $((s*8)) 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])')
    echo "blocks=$b in_flight=$s pairs/s=$v"
  done
done
