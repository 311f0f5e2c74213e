set -e
OUT=gpurun_out/profiles_out
mkdir -p $OUT
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
bash tools/pmc.sh r03_sq1 10m "$SQ1" 5 > $OUT/r03_10m_sq1.txt 2>&1
bash tools/pmc.sh r03_sq2 10m "$SQ2" 5 > $OUT/r03_10m_sq2.txt 2>&1
echo sq done
rm -f gpurun_out/r03q/bands3.log
for wl in 1m 10m 10m-4k 50m; do for rk in 1 2 4 8; do
python tools/band_bench.py --workload $wl --ranks $rk 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print(d['workload'],d['ranks'],d['band'],'ms',d['ms_per_frame'],'V',d['visible'],'D',d['pairs'],{k:s[k] for k in ('preprocess','depth_sort','expand','tile_sort','blend')})" | tee -a gpurun_out/r03q/bands3.log
done; done
