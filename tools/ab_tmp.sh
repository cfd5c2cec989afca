mkdir -p gpurun_out/ab
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --cpu-steps 1 --cpu-warmup 0 --cpu-steps-1thread 0 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1])
r=d['roofline']; k=r['other_kernels']
print('$name:', round(d['ms_per_step'],3), 'native', round(d['native_fp32']['ms_per_step'],2), 'attn fwd/bwd', k['glowtts_rel_attn_fwd_ex']['mean_us'], k['glowtts_rel_attn_bwd_ex']['mean_us'])"
}
for rep in 1 2; do
run new_$rep A=1
done
