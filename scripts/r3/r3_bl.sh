#!/bin/bash
# slow PROCESS or slow STEPS?  ten processes of 40 synchronised steps each at batch 512, GPU clocks before / after
OUT=gpurun_out/r3bl; mkdir -p $OUT
(rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4) | tee -a $OUT/probe.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 200 python -u scripts/slowmode_probe.py 512 40 2>&1 | grep -v amdgpu | tee -a $OUT/probe.txt
done
(rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4) | tee -a $OUT/probe.txt
