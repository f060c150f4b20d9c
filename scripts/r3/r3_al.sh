#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3al
timeout -k 10 200 python -u scripts/host_bound_gpt2.py 1 2>&1 | grep -v amdgpu | tee gpurun_out/r3al/host.txt
timeout -k 10 200 python -u scripts/host_bound_gpt2.py 2 2>&1 | grep -v amdgpu | tee -a gpurun_out/r3al/host.txt
timeout -k 10 200 python -u scripts/host_bound_gpt2.py 16 2>&1 | grep -v amdgpu | tee -a gpurun_out/r3al/host.txt
