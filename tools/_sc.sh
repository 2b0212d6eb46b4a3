cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for fs in 0 1; do echo "== VP_FOLD_STATS=$fs"; VP_FOLD_STATS=$fs timeout 300 python bench.py --variant l --dataset coco_25 --batch 8 --steps 50 --warmup 10 --no-cpu-baseline --no-host-path 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'): d=json.loads(l); print(d['value'], d['ms_per_step'], d['step_ms'])
"; done
VP_FOLD_STATS=1 timeout 300 python tools/stream_bench.py --frames 150 --persons 8 2>&1 | tail -1 | cut -c1-400
python tools/determinism_check.py s 16 2 | tail -1
