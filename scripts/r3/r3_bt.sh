#!/bin/bash
# soak on the final tree (paced host, default batches): long runs must hold the short runs' throughput
OUT=gpurun_out/r3bt; mkdir -p $OUT
line() { grep '^{' $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$2:', d['value'], d['unit'], d['ms_per_step'], 'ms loss', round(d['final_loss'], 5), 'reserved GB', c['max_memory_reserved_gb'])" | tee -a $OUT/soak.txt; }
timeout -k 10 500 python bench.py --steps 600 --warmup 10 > $OUT/r50.txt 2>&1 || { tail -5 $OUT/r50.txt; exit 1; }
line $OUT/r50.txt "resnet50 b1024 600 steps"
timeout -k 10 400 python bench.py --model gpt2 --steps 1500 --warmup 10 > $OUT/gpt2.txt 2>&1 || { tail -5 $OUT/gpt2.txt; exit 1; }
line $OUT/gpt2.txt "gpt2 16x1024 1500 steps"
timeout -k 10 500 python bench.py --model resnet152 --steps 200 --warmup 5 > $OUT/r152.txt 2>&1 || { tail -5 $OUT/r152.txt; exit 1; }
line $OUT/r152.txt "resnet152 b1024 200 steps"
timeout -k 10 300 python bench.py --model mnist --steps 30000 --warmup 100 > $OUT/mnist.txt 2>&1 || { tail -5 $OUT/mnist.txt; exit 1; }
line $OUT/mnist.txt "mnist 30000 steps"
