for m in vrnn srnn lstm cwvae wavenet stcn; do
  for d in f32 bf16; do
    timeout -k 10 280 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --dtype $d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m','$d', round(r['ms_per_step'],2),'ms/step', '%.3g'%r['value'],'frames/s bpd',r['bits_per_dim'])"
  done
done
