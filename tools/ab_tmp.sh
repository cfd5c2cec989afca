mkdir -p gpurun_out/ab
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --cpu-steps 1 --cpu-warmup 0 --cpu-steps-1thread 0 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1])
r=d['roofline']; k=r['mfma_kernels']
pick=['glowtts_conv_wrw[M384 K192x5 N32x400]','glowtts_conv_gate_fwd[M384 K192x5 N32x400]','glowtts_conv_fwd[M192 K384x5 N32x400]','glowtts_conv_gate_bwd[M192 K384x1 N32x400]','glowtts_conv_res_skip_fwd[M384 K192x1 N32x400]','glowtts_conv_wrw[M384 K192x1 N32x400]']
print('$name:', round(d['ms_per_step'],3), 'native', round(d['native_fp32']['ms_per_step'],2), ' '.join(str(k[x]['mean_us']) for x in pick))"
}
for rep in 1 2; do
run new_$rep A=1
done
