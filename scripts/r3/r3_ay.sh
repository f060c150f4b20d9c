#!/bin/bash
# soak on the end-of-round code: 1000-step runs (throughput must match the 20-step figure, loss finite, memory flat)
OUT=gpurun_out/r3ay; mkdir -p $OUT
timeout -k 10 400 python bench.py --steps 1000 --warmup 10 > $OUT/r50.txt 2>&1 || { tail -5 $OUT/r50.txt; exit 1; }
grep '^{' $OUT/r50.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet50 1000 steps:', d['value'], d['unit'], d['ms_per_step'], 'ms loss', d['final_loss'])" | tee -a $OUT/soak.txt
timeout -k 10 400 python bench.py --model gpt2 --steps 1000 --warmup 10 > $OUT/gpt2.txt 2>&1 || { tail -5 $OUT/gpt2.txt; exit 1; }
grep '^{' $OUT/gpt2.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gpt2 1000 steps:', d['value'], d['unit'], d['ms_per_step'], 'ms loss', d['final_loss'])" | tee -a $OUT/soak.txt
timeout -k 10 400 python bench.py --model resnet152 --steps 300 --warmup 5 > $OUT/r152.txt 2>&1 || { tail -5 $OUT/r152.txt; exit 1; }
grep '^{' $OUT/r152.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet152 300 steps:', d['value'], d['unit'], d['ms_per_step'], 'ms loss', d['final_loss'])" | tee -a $OUT/soak.txt
timeout -k 10 300 python bench.py --model mnist --steps 20000 --warmup 100 > $OUT/mnist.txt 2>&1 || { tail -5 $OUT/mnist.txt; exit 1; }
grep '^{' $OUT/mnist.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mnist 20000 steps:', d['value'], d['unit'], d['ms_per_step'], 'ms loss', d['final_loss'])" | tee -a $OUT/soak.txt
