# L2 (TCC) request counters per conv kernel: bash tools/pmc_l2_quick.sh <tag>
TAG=${1:-l2}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/$TAG/avail.txt 2>&1
for set in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_BUSY_sum TCC_TAG_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$TAG/$tag -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$TAG/$tag.log 2>&1
done
cd $R
python - <<PY
import sys, os
sys.path.insert(0, 'tools')
import summarize_rocprof as sr
base = 'gpurun_out/$TAG'
for sub in sorted(os.listdir(base)):
    d = os.path.join(base, sub)
    if not os.path.isdir(d): continue
    fs = [os.path.join(dp, x) for dp, _, xs in os.walk(d) for x in xs if x.endswith('counter_collection.csv')]
    if not fs: print(sub, 'no csv'); continue
    import csv
    names = set()
    with open(fs[0]) as f:
        for row in csv.DictReader(f): names.add(row.get('Counter_Name'))
    for k, v in sorted(sr.per_kernel(fs[0], names).items()):
        if 'igemm' not in k and 'wino' not in k: continue
        print('{:<42}'.format(k), ' '.join('{}={:.4g}'.format(n, v[n]) for n in sorted(names) if n in v))
PY
