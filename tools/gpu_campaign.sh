# One gpurun call that refreshes the evidence under profiles/: tests, default bench, rocprofv3 kernel stats and the three
# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters, each in a run of its own with --kernel-trace only).  The raw CSVs
# stay in /tmp on the box; only the condensed files land in gpurun_out/$TAG/ (merged back), to be copied into profiles/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_campaign.sh r01'
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
P=/tmp/prof_$TAG
mkdir -p $OUT $P
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json   # the driver's flags
python bench.py --no-ws --no-cpu-baseline > $OUT/bench_no_ws.json 2> $OUT/bench_no_ws.err; cut -c1-200 $OUT/bench_no_ws.json
python bench.py --workload isic > $OUT/bench_isic.json 2> $OUT/bench_isic.err; cut -c1-200 $OUT/bench_isic.json
# round 6: the shapes of the reference's real data (padded levels): 155 slices of 4x240x240, 32 images of 3x192x256 -- the first is also the default line's `native_shapes` sub-record
python bench.py --workload brats-native --steps 10 --warmup 3 > $OUT/bench_brats_native.json 2> $OUT/bench_brats_native.err; cut -c1-200 $OUT/bench_brats_native.json
python bench.py --workload isic-native > $OUT/bench_isic_native.json 2> $OUT/bench_isic_native.err; cut -c1-200 $OUT/bench_isic_native.json
python bench.py --ensemble 10 > $OUT/bench_ensemble10.json 2> $OUT/bench_ensemble10.err; cut -c1-200 $OUT/bench_ensemble10.json
python bench.py --aleatoric --mc 50 --steps 4 --no-cpu-baseline > $OUT/bench_aleatoric_mc50.json 2> $OUT/bench_aleatoric_mc50.err; cut -c1-200 $OUT/bench_aleatoric_mc50.json
# every output in the timed region (mutual information + variance: float64 statistics); the default line carries the same configuration as its `all_outputs` sub-record
python bench.py --all-outputs --steps 20 --warmup 5 > $OUT/bench_all_outputs.json 2> $OUT/bench_all_outputs.err; cut -c1-200 $OUT/bench_all_outputs.json
# the N > 1 exchange path over RCCL itself, as far as one GPU allows (a one-rank process group), and the bench line through it
for m in lazy; do python tools/rccl_world1_rehearsal.py 160 20 8 $m 2> $OUT/rccl_world1.err | grep '^{"backend"' >> $OUT/rccl_world1.json; done; cut -c1-300 $OUT/rccl_world1.json
RCU_BENCH_FORCE_PG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $OUT/bench_force_pg.err | grep '^{"metric"' > $OUT/bench_force_pg.json; cut -c1-200 $OUT/bench_force_pg.json
# (round 5: the eager-communicator leg, the RCCL kernel trace, the wait / instruction-mix / aggregation counter passes measure things that did not change: profiles/r04_*)
# the N = 8 lines with all eight ranks on the one GPU over gloo (code-path runs, not throughputs): MC and -- now that members share a workspace -- the K = 10 ensemble
# (--lanes 1: since round 5 every rank sizes its plans for the canonical 640-sample launch on every lane -- 24.4 GB per lane; eight ranks with two lanes each do not fit ONE GPU's 288 GB, which only this rehearsal asks of them)
RCU_BENCH_SINGLE_DEVICE=1 RCU_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline 2> $OUT/bench_gpus8_one_device.err | grep '^{"metric"' > $OUT/bench_gpus8_one_device.json; cut -c1-200 $OUT/bench_gpus8_one_device.json
RCU_BENCH_SINGLE_DEVICE=1 RCU_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --ensemble 10 --steps 8 --warmup 2 --no-cpu-baseline 2> $OUT/bench_ens10_gpus8_one_device.err | grep '^{"metric"' > $OUT/bench_ens10_gpus8_one_device.json; cut -c1-200 $OUT/bench_ens10_gpus8_one_device.json
cd /tmp && export TMPDIR=/tmp
# kernel times: one lane (exclusive durations, what bench.py's roofline record is taken from); then the default two lanes, whose kernels overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 > $P/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats2 -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $P/stats2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_sq.log 2>&1
# round 6: the same for the native BraTS shape (one lane) and the direct-kernel family (tools/fallback_profile.py: 240x240 under pad_levels = 0, and 100x100 with its centre pads)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_native -o bench -- python3 $R/bench.py --workload brats-native --brief --steps 2 --warmup 1 --no-cpu-baseline --lanes 1 > $P/stats_native.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch_native -o bench -- python3 $R/bench.py --workload brats-native --brief --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_fetch_native.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write_native -o bench -- python3 $R/bench.py --workload brats-native --brief --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_write_native.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq_native -o bench -- python3 $R/bench.py --workload brats-native --brief --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/pmc_sq_native.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_fallback -o fb -- python3 $R/tools/fallback_profile.py 3 > $P/stats_fallback.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch_fallback -o fb -- python3 $R/tools/fallback_profile.py 1 > $P/pmc_fetch_fallback.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write_fallback -o fb -- python3 $R/tools/fallback_profile.py 1 > $P/pmc_write_fallback.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq_fallback -o fb -- python3 $R/tools/fallback_profile.py 1 > $P/pmc_sq_fallback.log 2>&1
cd $R
python tools/summarize_rocprof.py stats $(find $P/stats_native -name "*kernel_stats.csv" | head -1) $OUT/bench_brats_native_steps2_kernel_stats.csv
python tools/summarize_rocprof.py stats $(find $P/stats_fallback -name "*kernel_stats.csv" | head -1) $OUT/fallback_kernel_stats.csv
python tools/summarize_rocprof.py pmc $P/pmc_fetch_native $P/pmc_write_native $P/pmc_sq_native $OUT/bench_brats_native_pmc.json --plan $OUT/bench_brats_native.json --traffic-name pmc_traffic_brats-native.json
python tools/summarize_rocprof.py pmc $P/pmc_fetch_fallback $P/pmc_write_fallback $P/pmc_sq_fallback $OUT/fallback_pmc.json --traffic-name fallback_traffic.json
python tools/fallback_profile.py 5 > $OUT/fallback_layer_report.txt 2>&1
python tools/summarize_rocprof.py stats $(find $P/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_steps2_kernel_stats.csv
python tools/summarize_rocprof.py stats $(find $P/stats2 -name "*kernel_stats.csv" | head -1) $OUT/bench_steps2_lanes2_kernel_stats.csv
python bench.py --lanes 1 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_lanes1.json 2> $OUT/bench_lanes1.err; cut -c1-200 $OUT/bench_lanes1.json
python tools/summarize_rocprof.py pmc $P/pmc_fetch $P/pmc_write $P/pmc_sq $OUT/bench_pmc.json --plan $OUT/bench_default.json
head -12 $OUT/bench_steps2_kernel_stats.csv
timeout 200 python tools/calib_bench.py 160 > $OUT/calib_bench.json 2>/dev/null
python __graft_entry__.py smoke 2>&1 | tail -1 > $OUT/smoke.txt; cat $OUT/smoke.txt
# round 5: the compute side of 8 GPUs on one (rank 0's exact job list), and the evaluation script's hot loop end to end
timeout 300 python tools/rank_share_of_world.py --out $OUT/rank_share_of_8.json > /dev/null 2> $OUT/rank_share.err
timeout 600 python tools/eval_throughput.py --subjects 32 --out $OUT/eval_throughput.json > /dev/null 2> $OUT/eval_throughput.err
timeout 300 python tools/layer_report.py 5 640 > $OUT/layer_report_640.txt 2>&1
# round 6: the reference's real shapes, padded levels against the plans of rounds 1-5 (pad_levels=0)
timeout 300 python tools/layer_report.py 5 155 shape=240x240 > $OUT/layer_report_native_155.txt 2>&1
timeout 300 python tools/layer_report.py 5 155 shape=240x240 pad_levels=0 > $OUT/layer_report_native_155_nopad.txt 2>&1
timeout 300 python tools/layer_report.py 5 32 shape=192x256 cin=3 > $OUT/layer_report_isic_192x256_32.txt 2>&1
timeout 300 python tools/layer_report.py 5 32 shape=192x256 cin=3 pad_levels=0 > $OUT/layer_report_isic_192x256_32_nopad.txt 2>&1
timeout 100 tools/microbench/bf16x3_chunk_bench > $OUT/bf16x3_chunk_bench.txt 2>&1
RCU_SCRIPT_PROFILE=0 timeout 200 python tools/script_throughput.py 16 20 32 > /dev/null 2>&1    # (a first run on a fresh box pages the interpreter, the NIfTI writers' zlib ... in: 0.12-0.16 s per subject; the recorded run is the second)
RCU_SCRIPT_PROFILE=0 timeout 200 python tools/script_throughput.py 16 20 32 3932160 timing 2>&1 | grep -v "Holder\|conv2d_batch\|Conv2d\|Dropout2d\|BatchNorm2d\|^ *)\|^UNet\|^model" > $OUT/script_throughput.txt
# round 6: the shipped YAML as it is (batch_size 32; coalescing is the scripts' default now) over 64 subjects, the K = 10 ensemble script, the other batch sizes, and the reference's real BraTS shape
RCU_SCRIPT_PROFILE=0 timeout 400 python tools/script_throughput.py 64 20 32 2>&1 | grep -v "Holder\|conv2d_batch\|Conv2d\|Dropout2d\|BatchNorm2d\|^ *)\|^UNet\|^model" | tail -4 > $OUT/script_throughput_64.txt
RCU_SCRIPT_ENSEMBLE=10 RCU_SCRIPT_PROFILE=0 timeout 400 python tools/script_throughput.py 32 20 32 2>&1 | grep -v "Holder\|conv2d_batch\|Conv2d\|Dropout2d\|BatchNorm2d\|^ *)\|^UNet\|^model" | tail -4 > $OUT/script_throughput_ensemble.txt
RCU_SCRIPT_PROFILE=0 timeout 400 python tools/script_throughput.py 32 20 8 2>&1 | tail -3 > $OUT/script_throughput_batch8.txt
RCU_SCRIPT_PROFILE=0 timeout 400 python tools/script_throughput.py 32 20 160 2>&1 | tail -3 > $OUT/script_throughput_batch160.txt
RCU_SCRIPT_NATIVE=1 RCU_SCRIPT_PROFILE=0 timeout 600 python tools/script_throughput.py 24 20 32 2>&1 | tail -3 > $OUT/script_throughput_native.txt
timeout 300 python tools/isic_script_throughput.py > $OUT/isic_script_throughput.txt 2>&1
# the shipped batch_size: 32 as it is (no coalescing): the loop's run-ahead (rcu_amd.loops.Test.INFLIGHT_PIXELS) is what keeps the GPU busy there
RCU_SCRIPT_PROFILE=0 timeout 200 python tools/script_throughput.py 16 20 32 0 timing 2>&1 | grep -v "Holder\|conv2d_batch\|Conv2d\|Dropout2d\|BatchNorm2d\|^ *)\|^UNet\|^model" > $OUT/script_throughput_batch32.txt
ls -la $OUT
