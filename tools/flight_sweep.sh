#!/bin/bash
# frames in flight: the headline workload (and the larger ones) with 1 / 2 / 3 renderers on priority streams
# usage: tools/flight_sweep.sh "1m 10m 50m" "1 2 3"
for wl in ${1:-"1m 10m"}; do
  for f in ${2:-"1 2 3"}; do
    python3 bench.py --workload $wl --no-roofline --no-cpu-baseline --extra-workloads "" --steps 20 --warmup 5 --timing-steps 0 \
      --frame-samples 0 --frames-in-flight $f 2>/dev/null | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$wl frames in flight $f: ms_per_step %.4f  single %s  steady %s' % (j['ms_per_step'], (j.get('single_stream') or {}).get('ms_per_step'), (j.get('steady_state') or {}).get('ms_per_step')))"
  done
done
