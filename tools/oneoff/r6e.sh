mkdir -p gpurun_out/r6e
(timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6e/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6e/gputests.log)
(timeout 300 python -m pytest tests/test_gpu_production_sizes.py -k content -q -s -p no:cacheprovider 2>&1 | grep "^\[content\|passed\|failed" > gpurun_out/r6e/content_lines.txt)
VP_PROBE_SET=operand_bits timeout 300 python tools/clock_power_probe.py --secs 4 > gpurun_out/r6e/operand_bits.txt 2>&1
timeout 300 python bench.py > gpurun_out/r6e/bench.json 2> gpurun_out/r6e/bench.err
tail -3 gpurun_out/r6e/gputests.log; cat gpurun_out/r6e/content_lines.txt | cut -c1-300; cat gpurun_out/r6e/operand_bits.txt | tail -12; head -c 400 gpurun_out/r6e/bench.json
