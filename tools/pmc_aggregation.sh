# What bounds the standalone aggregation kernels (head_kernel / head_stream_kernel, mc_accumulate, mc_finalize): SQ wave-cycle breakdown, occupancy and
# L2 (TCC) hit / miss counters, each counter set in a run of its own (--kernel-trace only).   bash tools/pmc_aggregation.sh <tag>
TAG=${1:-aggpmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM SQ_INSTS_VALU"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
P5="FETCH_SIZE"
P6="WRITE_SIZE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/$TAG/p$i -o agg -- python3 $R/tools/agg_bench.py 3 > $R/gpurun_out/$TAG/p$i.log 2>&1
  i=$((i+1))
done
cd $R
python - <<PY
import sys, os, collections, csv
sys.path.insert(0, 'tools')
import summarize_rocprof as sr
tot = collections.defaultdict(dict)
for i in range(1, 7):
    d = 'gpurun_out/$TAG/p%d' % i
    fs = [os.path.join(dp, x) for dp, _, xs in os.walk(d) for x in xs if x.endswith('counter_collection.csv')]
    if not fs:
        print('pass', i, 'produced no counters'); continue
    names = set(r['Counter_Name'] for r in csv.DictReader(open(fs[0])))
    for k, v in sr.per_kernel(fs[0], names).items():
        tot[k].update(v)
print('# per launch, averaged over the launches of tools/agg_bench.py (160- and 640-slice launches, float32 and float64 statistics mixed); counters of separate runs')
for k, v in sorted(tot.items()):
    if not any(t in k for t in ('head', 'mc_accumulate', 'mc_finalize')): continue
    g = lambda n: v.get(n, float('nan'))
    wc, gui = g('SQ_WAVE_CYCLES'), g('GRBM_GUI_ACTIVE') / 8
    print(k)
    print('   waves {:.0f}; wave cycles per SIMD-cycle (occupancy in waves per SIMD) {:.2f}; CU busy {:.2f}'.format(g('SQ_WAVES'), wc / 1024 / gui, g('SQ_BUSY_CU_CYCLES') / 256 / gui if g('SQ_BUSY_CU_CYCLES') == g('SQ_BUSY_CU_CYCLES') else float('nan')))
    print('   of wave cycles: wait_any {:.3f} wait_inst_any {:.3f} active_inst_any {:.3f} | valu {:.3f} vmem {:.3f} lds {:.3f} scalar {:.3f}'.format(
        g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_ACTIVE_INST_VALU') / wc, g('SQ_ACTIVE_INST_VMEM') / wc,
        g('SQ_ACTIVE_INST_LDS') / wc, g('SQ_ACTIVE_INST_SCA') / wc))
    print('   instructions per wave: vmem {:.1f} valu {:.1f}'.format(g('SQ_INSTS_VMEM') / g('SQ_WAVES'), g('SQ_INSTS_VALU') / g('SQ_WAVES')))
    hit, miss = g('TCC_HIT_sum'), g('TCC_MISS_sum')
    print('   L2: hit {:.3g} miss {:.3g} (hit rate {:.3f}); EA read requests {:.3g} write requests {:.3g}; tag stall cycles / GUI cycle {:.3f}; TCP pending stall / GUI cycle {:.3f}'.format(
        hit, miss, hit / max(hit + miss, 1), g('TCC_EA0_RDREQ_sum'), g('TCC_EA0_WRREQ_sum'), g('TCC_TAG_STALL_sum') / gui, g('TCP_PENDING_STALL_CYCLES_sum') / gui))
    print('   HBM bytes (2 FETCH_SIZE + WRITE_SIZE) x 1024: {:.4g}'.format((2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024))
PY
