# The rocprofv3 legs of tools/gpu_campaign.sh alone (kernel stats on one and two lanes, the three PMC passes) plus the 8-ranks-on-one-GPU rehearsal:
#   gpurun --timeout 1200 -- 'bash tools/gpu_campaign_profiles.sh r05 profiles/r05_bench_default.json'
TAG=${1:-r05}
PLAN=${2:-profiles/${TAG}_bench_default.json}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
P=/tmp/prof_$TAG
mkdir -p $OUT $P
RCU_BENCH_SINGLE_DEVICE=1 RCU_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline 2> $OUT/bench_gpus8_one_device.err | grep '^{"metric"' > $OUT/bench_gpus8_one_device.json; cut -c1-200 $OUT/bench_gpus8_one_device.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 > $P/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats2 -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $P/stats2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_sq.log 2>&1
cd $R
python tools/summarize_rocprof.py stats $(find $P/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_steps2_kernel_stats.csv
python tools/summarize_rocprof.py stats $(find $P/stats2 -name "*kernel_stats.csv" | head -1) $OUT/bench_steps2_lanes2_kernel_stats.csv
python tools/summarize_rocprof.py pmc $P/pmc_fetch $P/pmc_write $P/pmc_sq $OUT/bench_pmc.json --plan $PLAN
head -14 $OUT/bench_steps2_kernel_stats.csv
cat $OUT/pmc_traffic.json
