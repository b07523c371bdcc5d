OUT=$GRAFT_REPO_ROOT/gpurun_out/r04l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export RCU_HIP_LIBRARY=$GRAFT_REPO_ROOT/reliability-challenges-uncertainty_amd/librcu_hip_exp.so
RCU_EXP_VERBOSE=1 RCU_EXP_STAGGER=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "^tensor" | head -40 > $OUT/addr.txt; head -12 $OUT/addr.txt
for rep in 1 2; do
for S in 0 4096 69632 266240 1052672 0; do
  RCU_EXP_STAGGER=$S python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/s_${S}_$rep.json
  python - $OUT/s_${S}_$rep.json $S <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('stagger %8s value %.2f resident %.2f conv %.3f ms  T32x32 %.4f  S2 %.4f S8 %.4f first %.4f up32 %.4f cls %.4f'%(sys.argv[2],d['value'],d['resident']['value'],r['all_conv_kernels']['ms_per_forward'],
   r['per_kernel']['conv3x3_winograd4<T32x32,N32,K8>']['ms_per_forward'],r['per_kernel']['conv3x3_winograd4<S2T16x32,N32,K8>']['ms_per_forward'],r['per_kernel']['conv3x3_winograd4<S8T8x16,N32,K8>']['ms_per_forward'],
   r['per_kernel']['conv3x3_first<T8x32,K36>']['ms_per_forward'],r['per_kernel']['upconv_winograd<T16x32,N32,K8>']['ms_per_forward'],r['per_kernel']['conv3x3_winograd<T16x32,N32,K8>']['ms_per_forward']))
PY
done; done
