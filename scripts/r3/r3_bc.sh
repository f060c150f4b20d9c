#!/bin/bash
mkdir -p gpurun_out/r3bc
bash scripts/rehearse_multirank.sh > gpurun_out/r3bc/rehearse.txt 2>&1; rc=$?
cat gpurun_out/r3bc/rehearse.txt | cut -c1-220
exit $rc
