timeout 200 python bench.py --variant l --dataset coco_25 --batch 8 --input u8 --steps 200 --warmup 20 --no-cpu-baseline --no-host-path --breakdown 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{\"breakdown\"'):
        b=json.loads(l)['breakdown']
        tot=0
        for k,v in b.items(): print(k, v); tot+=v['ms_per_step']
        print('sum of kernel ms', tot)
    elif l.startswith('{\"metric\"'):
        j=json.loads(l); print('value', j['value'], 'ms_per_step', j['ms_per_step'], j['step_ms'])
"
