OUT=$GRAFT_REPO_ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in lazy eager lazy eager; do python tools/rccl_world1_rehearsal.py 160 20 8 $m 2>/dev/null | grep '^{"backend"' | tee -a $OUT/rehearsal.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['init'], 'plain %.2f / with copy %.2f | reduce %.2f / %.2f | p2p %.2f / %.2f | bits %s'%(d['plain_ms_per_volume'],d['plain_ms_per_volume_with_copy'],d['reduce']['ms_per_volume'],d['reduce']['ms_per_volume_with_copy'],d['p2p']['ms_per_volume'],d['p2p']['ms_per_volume_with_copy'],d['bits_equal']))"; done
RCU_BENCH_FORCE_PG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/bench_force_pg.json; cut -c1-150 $OUT/bench_force_pg.json
RCU_BENCH_FORCE_PG=1 RCU_BENCH_PG_EAGER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/bench_force_pg_eager.json; cut -c1-150 $OUT/bench_force_pg_eager.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/bench_plain.json; cut -c1-150 $OUT/bench_plain.json
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3
