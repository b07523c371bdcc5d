# HBM bytes per launch of the calibration kernels (FETCH_SIZE / WRITE_SIZE, separate passes with --kernel-trace only) on tools/calib_bench.py's 160-volume batch:
#   gpurun --timeout 600 -- 'bash tools/pmc_calibration.sh r05_calibpmc'
TAG=${1:-calibpmc}
R=$GRAFT_REPO_ROOT
P=/tmp/$TAG
mkdir -p $R/gpurun_out/$TAG $P
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/$c -o calib -- python3 $R/tools/calib_bench.py 160 > $P/$c.log 2>&1
done
cd $R
python - <<PY
import csv, collections, glob
out = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    path = glob.glob('$P/%s/**/*counter_collection.csv' % c, recursive=True)[0]
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == c and ('ece_hist' in r['Kernel_Name'] or 'unc_counts' in r['Kernel_Name']):
            per[(r['Kernel_Name'][:110], r['Dispatch_Id'])] += float(r['Counter_Value'])
    for (k, _), v in per.items():
        out[k][c].append(v)
n = 160 * 160 * 192 * 128
print('# per launch over the 160-volume batch (%d voxels); HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts half of wide reads); largest launches only' % n)
for k, d in sorted(out.items()):
    f = max(d['FETCH_SIZE']) if d['FETCH_SIZE'] else 0
    w = max(d['WRITE_SIZE']) if d['WRITE_SIZE'] else 0
    print('%-112s launches %3d  bytes/voxel %.3f  (fetch %.0f KiB, write %.0f KiB)' % (k, len(d['FETCH_SIZE']), (2 * f + w) * 1024 / n, f, w))
PY
