# Instruction mix of the conv kernels per MFMA (SQ instruction counters): bash tools/pmc_inst_mix.sh <tag>
TAG=${1:-mix}
R=$GRAFT_REPO_ROOT
P=/tmp/prof_$TAG
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $P/p1 -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $P/p1.log 2>&1
cd $R
python - <<PY
import sys, os
sys.path.insert(0, 'tools')
import summarize_rocprof as sr
d = '$P/p1'
f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith('counter_collection.csv')][0]
names = {'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_WAVES'}
for k, v in sorted(sr.per_kernel(f, names).items()):
    if 'igemm' not in k and 'wino' not in k: continue
    m = max(v.get('SQ_INSTS_MFMA', 0), 1)
    print('{:<42} per MFMA: valu(non-mfma) {:.2f} salu {:.2f} lds {:.2f} vmem {:.3f} smem {:.3f}'.format(
        k, (v['SQ_INSTS_VALU'] - v.get('SQ_INSTS_MFMA', 0)) / m, v['SQ_INSTS_SALU'] / m, v['SQ_INSTS_LDS'] / m, v['SQ_INSTS_VMEM'] / m, v['SQ_INSTS_SMEM'] / m))
PY
