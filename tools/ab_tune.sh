for r in 1 2; do for t in "9:8:0" "0:11:0" "0:11:0,2:11:8" "0:11:0,2:11:8,1:11:8" "0:11:0,4:11:0"; do
  echo -n "TUNE=$t: "
  VP_GEMM_TUNE=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[:9]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
"; done; done
