#!/bin/bash
set -o pipefail
export GPU_MAX_HW_QUEUES=8
bash scripts/gpu_profile.sh gpt2_r3t --model gpt2 --batch 16 || exit 1
DPA_WGRAD_STREAM=0 bash scripts/gpu_profile.sh gpt2_r3t_alone --model gpt2 --batch 16 || exit 1
