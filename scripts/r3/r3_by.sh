#!/bin/bash
OUT=gpurun_out/r3by; mkdir -p $OUT
timeout -k 10 300 python -u scripts/host_profile.py gpt2 10 > $OUT/gpt2_host.txt 2>&1 || { tail -20 $OUT/gpt2_host.txt; exit 1; }
grep -A 40 "tottime" $OUT/gpt2_host.txt | cut -c1-150 | head -58
