# Where the conv kernels' wave cycles go (three counter passes): bash tools/pmc_wait_breakdown.sh <tag>
TAG=${1:-waits}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES"
P3="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_INSTS_MFMA SQ_INSTS_VMEM"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/$TAG/p$i -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $R/gpurun_out/$TAG/p$i.log 2>&1
  i=$((i+1))
done
cd $R
python - <<PY
import sys, os, collections
sys.path.insert(0, 'tools')
import summarize_rocprof as sr
tot = collections.defaultdict(dict)
for i in (1, 2, 3):
    d = 'gpurun_out/$TAG/p%d' % i
    f = [os.path.join(d, x) for x in os.listdir(d) if x.endswith('counter_collection.csv')][0]
    import csv
    names = set(r['Counter_Name'] for r in csv.DictReader(open(f)))
    for k, v in sr.per_kernel(f, names).items():
        tot[k].update(v)
for k, v in sorted(tot.items()):
    if 'igemm' not in k and 'wino' not in k: continue
    wc = v['SQ_WAVE_CYCLES']
    print(k)
    print('   of wave cycles: wait_any {:.3f} wait_inst_any {:.3f} (lds {:.3f}) active_inst_any {:.3f}'.format(
        v['SQ_WAIT_ANY'] / wc, v['SQ_WAIT_INST_ANY'] / wc, v['SQ_WAIT_INST_LDS'] / wc, v['SQ_ACTIVE_INST_ANY'] / wc))
    print('   active: valu {:.3f} vmem {:.3f} lds {:.3f} misc {:.3f} scalar {:.3f};  vmem_rd cycles {:.3f} vmem_wr {:.3f}'.format(
        v['SQ_ACTIVE_INST_VALU'] / wc, v['SQ_ACTIVE_INST_VMEM'] / wc, v['SQ_ACTIVE_INST_LDS'] / wc, v['SQ_ACTIVE_INST_MISC'] / wc,
        v['SQ_ACTIVE_INST_SCA'] / wc, v['SQ_INST_CYCLES_VMEM_RD'] / wc, v['SQ_INST_CYCLES_VMEM_WR'] / wc))
    gui = v['GRBM_GUI_ACTIVE'] / 8
    print('   per SIMD-cycle: mfma busy {:.3f} coexec {:.3f}; fifo full: lds_cmd {:.4f} lds_data {:.4f} ta_addr {:.4f} ta_cmd {:.4f} wr_data {:.4f}; ifetch/inst {:.3f}'.format(
        v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui, v['SQ_VALU_MFMA_COEXEC_CYCLES'] / 1024 / gui,
        v['SQ_LDS_CMD_FIFO_FULL'] / 1024 / gui, v['SQ_LDS_DATA_FIFO_FULL'] / 1024 / gui, v['SQ_VMEM_TA_ADDR_FIFO_FULL'] / 1024 / gui,
        v['SQ_VMEM_TA_CMD_FIFO_FULL'] / 1024 / gui, v['SQ_VMEM_WR_TA_DATA_FIFO_FULL'] / 1024 / gui,
        v['SQ_IFETCH'] / max(v['SQ_INSTS_MFMA'] + v['SQ_INSTS_VMEM'], 1)))
PY
