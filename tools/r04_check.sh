OUT=$GRAFT_REPO_ROOT/gpurun_out/r04m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/reliability-challenges-uncertainty_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "pass_pair or mc20 or 160_slice" 2>&1 | tail -3
for i in 1 2 3; do
RCU_HIP_LIBRARY=$L/librcu_hip_prev.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/prev_$i.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/new_$i.json
python - $OUT $i <<'PY'
import json,sys
o,i=sys.argv[1],sys.argv[2]
a=json.load(open('%s/prev_%s.json'%(o,i))); b=json.load(open('%s/new_%s.json'%(o,i)))
k='conv3x3_winograd<T16x32,N32,K8>'
print(i,'prev %.2f (conv %.3f ms, +head %.4f)   new %.2f (conv %.3f ms, +head %.4f)  all_out prev %.2f new %.2f'%(a['value'],a['roofline']['all_conv_kernels']['ms_per_forward'],a['roofline']['per_kernel'][k]['ms_per_forward'],b['value'],b['roofline']['all_conv_kernels']['ms_per_forward'],b['roofline']['per_kernel'][k]['ms_per_forward'],a['all_outputs']['value'],b['all_outputs']['value']))
PY
done
