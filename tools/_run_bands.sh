set -e
mkdir -p gpurun_out/r03q
rm -f gpurun_out/r03q/bands3.log
for wl in 1m 10m 10m-4k 50m; do for rk in 1 2 4 8; do
python tools/band_bench.py --workload $wl --ranks $rk 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print(d['workload'],d['ranks'],d['band'],'ms',d['ms_per_frame'],'V',d['visible'],'D',d['pairs'],{k:s[k] for k in ('preprocess','depth_sort','expand','tile_sort','blend')})" | tee -a gpurun_out/r03q/bands3.log
done; done
