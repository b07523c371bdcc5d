# quick HBM-traffic + timing check of the conv kernels: bash tools/pmc_quick.sh <tag>
TAG=${1:-quick}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
python tools/layer_report.py 5 2>&1 | tail -27 | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc_write.log 2>&1
cd $R
python tools/summarize_rocprof.py pmc gpurun_out/$TAG/pmc_fetch gpurun_out/$TAG/pmc_write gpurun_out/$TAG/pmc.json
python - <<PY
import json
d=json.load(open('gpurun_out/$TAG/pmc.json'))
for k,v in sorted(d.items()):
    print('{:<45} fetch {:>9.1f} MiB  write {:>9.1f} MiB  hbm {:>9.1f} MB'.format(k, v['fetch_size_kib_per_launch']/1024, v['write_size_kib_per_launch']/1024, v['hbm_bytes_per_launch']/1e6))
PY
