for rep in 1 2 3; do for st in 0 8192; do for w in chain5 tiler mosaic; do
  echo "stagger $st $w: $(GM_ALLOC_STAGGER=$st python bench.py --workload $w --steps 60 --warmup 10 --no-e2e --cpu-tiles 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['frac'], r['avg_launch_us'], r['kernel_only_us'])")"
done; done; done
