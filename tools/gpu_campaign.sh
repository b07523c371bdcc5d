set -x
mkdir -p gpurun_out/r01b
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r01b/bench_default.json 2> gpurun_out/r01b/bench_default.err; tail -c 3000 gpurun_out/r01b/bench_default.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/stats -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01b/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r01b/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01b/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r01b/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r01b/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r01b/pmc_sq.log 2>&1
cd $R
find gpurun_out/r01b -name "*.csv" | xargs ls -la
# keep only the needed CSVs small: drop huge traces
find gpurun_out/r01b -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r01b
