# LDS bank conflicts / MFMA busy per conv kernel: bash tools/pmc_sq_quick.sh <tag>
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc_sq.log 2>&1
cd $R
python - <<PY
import sys, os
sys.path.insert(0, 'tools')
import summarize_rocprof as sr
d = 'gpurun_out/$TAG/pmc_sq'
f = [os.path.join(d, x) for x in os.listdir(d) if x.endswith('counter_collection.csv')][0]
names = {'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'}
for k, v in sorted(sr.per_kernel(f, names).items()):
    if 'igemm' not in k and 'wino' not in k: continue
    print('{:<45} lds_conflict {:.3f}  mfma_busy {:.3f}  wait_any/wave_cycles {:.3f}'.format(
        k, v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1), v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (v['GRBM_GUI_ACTIVE'] / 8),
        v['SQ_WAIT_INST_ANY'] / max(v['SQ_WAVE_CYCLES'], 1)))
PY
