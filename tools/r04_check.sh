OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/reliability-challenges-uncertainty_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/t1.txt
for i in 1 2; do
for v in aggold headw4 headw16; do RCU_HIP_LIBRARY=$L/librcu_hip_$v.so python tools/agg_bench.py > $OUT/agg_${v}_$i.json 2>$OUT/agg_$v.err; done
python tools/agg_bench.py > $OUT/agg_new_$i.json 2>$OUT/agg_new.err
done
python - <<'PY'
import json,os
o=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04d/'
for i in (1,2):
    d={v: json.load(open(o+'agg_%s_%d.json'%(v,i))) for v in ('aggold','headw4','new','headw16')}
    for k in d['new']:
        if 'head' in k: print(i, '%-32s' % k, '  '.join('%s %6.1f us (%.2f)' % (v, d[v][k]['us'], d[v][k]['moved_frac']) for v in d))
PY
python bench.py --aleatoric --mc 50 --steps 4 --no-cpu-baseline > $OUT/bench_ale.json 2> $OUT/bench_ale.err; cut -c1-200 $OUT/bench_ale.json
RCU_HIP_LIBRARY=$L/librcu_hip_aggold.so python bench.py --aleatoric --mc 50 --steps 4 --no-cpu-baseline > $OUT/bench_ale_old.json 2> $OUT/bench_ale_old.err; cut -c1-200 $OUT/bench_ale_old.json
