#!/bin/bash
# the new default (per-GPU batch 1024) exactly as the driver runs it, three times, with peak memory; 512 beside it
OUT=gpurun_out/r3bh; mkdir -p $OUT
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | tail -1 >> $OUT/default.jsonl
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default:', d['value'], d['ms_per_step'], 'ms vs_baseline', d['vs_baseline'], 'batch', d['config']['per_gpu_batch'], 'mfu', d['mfu'])" | tee -a $OUT/ab.txt
  timeout -k 10 300 python bench.py --gpus 1 --batch 512 --steps 20 --warmup 5 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b512:', d['value'], d['ms_per_step'], 'ms vs_baseline', d['vs_baseline'])" | tee -a $OUT/ab.txt
done
python - <<'PY' | tee -a gpurun_out/r3bh/ab.txt
import torch, sys, os
sys.path.insert(0, os.getcwd())
from distributedpytorch_amd.models import resnet
from distributedpytorch_amd.parallel.dp import DataParallel
from distributedpytorch_amd import comm as dcomm
from distributedpytorch_amd.optim import FusedSGD
dev = torch.device("cuda", 0)
m = resnet.resnet50().to(dev).train()
x = torch.randn(1024, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (1024,), device=dev)
dp = DataParallel(m, dcomm.Communicator(device=dev)); opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)
for _ in range(3):
    m.loss(x, y).backward(); dp.finish(); opt.step()
torch.cuda.synchronize()
print("peak memory at batch 1024: %.1f GB allocated, %.1f GB reserved" % (torch.cuda.max_memory_allocated() / 1e9, torch.cuda.max_memory_reserved() / 1e9))
PY
