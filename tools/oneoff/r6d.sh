mkdir -p gpurun_out/r6d
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_api.py tests/test_gpu_production_sizes.py -k "query_split or split_k or content" -q -s -p no:cacheprovider > gpurun_out/r6d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6d/tests.log)
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-host-path --no-clock"
one() { "$@" 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for cfg in "--variant b --batch 1" "--variant l --dataset coco_25 --batch 1" "--variant b --batch 2" "--variant l --dataset coco_25 --batch 4" "--variant l --dataset coco_25 --batch 8" "--variant b --batch 8" "--variant h --dataset wholebody --batch 8" "--variant s --batch 1" "--variant b --batch 16"; do
  for q in 0 64 128 256 512; do echo -n "VP_ATTN_QSPLIT=$q $cfg: "; VP_ATTN_QSPLIT=$q one timeout 100 $B $cfg; done
  echo -n "VP_SPLITK=0 VP_ATTN_QSPLIT=0 $cfg: "; VP_SPLITK=0 VP_ATTN_QSPLIT=0 one timeout 100 $B $cfg
done > gpurun_out/r6d/qsplit.txt 2>&1
grep "^\[\|passed\|failed\|rc=" gpurun_out/r6d/tests.log | cut -c1-330; cat gpurun_out/r6d/qsplit.txt
