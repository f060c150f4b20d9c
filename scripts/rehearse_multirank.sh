#!/bin/bash
# Multi-rank flows rehearsed on ONE GPU: every rank on GPU 0 over gloo (RCCL refuses two ranks on one device), plus the
# RCCL call path at world size 1.  The numbers printed are NOT benchmark results.   scripts/rehearse_multirank.sh
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/rehearse
run() {  # name nproc args...
    local name=$1 n=$2; shift 2
    DPA_SHARE_GPU=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $((29500 + RANDOM % 400)) bench.py --gpus $n --backend gloo "$@" > gpurun_out/rehearse/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc $(grep '^{' gpurun_out/rehearse/$name.log | tail -1 | cut -c1-150)"
    [ $rc -eq 0 ] || { tail -15 gpurun_out/rehearse/$name.log; exit $rc; }
}
run dp2_r50 2 --batch 32 --steps 4 --warmup 2
run dp2_gpt2 2 --model gpt2 --batch 2 --steps 4 --warmup 2
run dp2_mnist 2 --model mnist --steps 50 --warmup 10
run dp2_r50_sharded 2 --batch 32 --steps 4 --warmup 2 --shard-optimizer
# gradient buckets through the one-kernel all-reduce over hipIpc-mapped peer memory (two ranks of one GPU map each other).
# Opt-in, like the tests of the same set-up (DPA_TEST_SHARED_GPU_PEER=1): whether two PROCESSES' kernels run side by side on
# one GPU is the driver's decision; when they do not, every bounded poll runs into its time limit before the layer declines.
if [ "${DPA_TEST_SHARED_GPU_PEER:-0}" = "1" ]; then
DPA_PEER_CHUNK_MB=8 DPA_PEER_TIMEOUT=5 run dp2_r50_ipc 2 --batch 32 --steps 4 --warmup 2 --allreduce ipc
grep -o '"allreduce": "[^"]*"' gpurun_out/rehearse/dp2_r50_ipc.log | tail -1
fi
# the same flows through the NATIVE layer's Python side (utils/fake_native.py: RcclComm's interface, data over gloo): self-test
# phases at world size > 1, buckets through _native_all_reduce, ZeRO-1 on native reduce / broadcast, and - for 4 and 8 ranks -
# what the driver's N = 4 / 8 runs do before the first step.  "comm_layer" must read native in these lines.
DPA_FAKE_NATIVE=1 run dp2_r50_nat 2 --batch 32 --steps 4 --warmup 2
grep -o '"comm_layer": "[^"]*"' gpurun_out/rehearse/dp2_r50_nat.log | tail -1
DPA_FAKE_NATIVE=1 run dp4_r50_nat 4 --batch 16 --steps 3 --warmup 1
DPA_FAKE_NATIVE=1 run dp8_mnist_nat 8 --model mnist --steps 30 --warmup 5
DPA_FAKE_NATIVE=1 run dp2_gpt2_nat 2 --model gpt2 --batch 2 --steps 4 --warmup 2
DPA_FAKE_NATIVE=1 run dp2_r50_sharded_nat 2 --batch 32 --steps 4 --warmup 2 --shard-optimizer
DPA_FAKE_NATIVE=1 run pp2_mnist_nat 2 --model mnist --pipeline 2 --steps 20 --warmup 5
DPA_FAKE_NATIVE=1 run pp2xdp2_mnist_nat 4 --model mnist --pipeline 2 --steps 10 --warmup 3
run pp2_mnist 2 --model mnist --pipeline 2 --steps 20 --warmup 5
run pp4_r152 4 --model resnet152 --pipeline 4 --batch 32 --steps 3 --warmup 1
# the RCCL path (native communicator, every bucket all-reduced for real) at world size 1
DPA_FORCE_COLLECTIVES=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/rehearse/rccl1_r50.log 2>&1 || { tail -15 gpurun_out/rehearse/rccl1_r50.log; exit 1; }
echo "rccl1_r50 $(grep '^{' gpurun_out/rehearse/rccl1_r50.log | tail -1 | cut -c1-150)"
grep -o '"comm_layer": "[^"]*"' gpurun_out/rehearse/rccl1_r50.log | tail -1
# one RCCL world communicator per process: with the native layer attached c10d is a gloo control plane, so the RCCL log
# shows ONE communicator initialisation (round 2 had two: c10d's ProcessGroupNCCL beside the native one)
DPA_FORCE_COLLECTIVES=1 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --model tiny_resnet --batch 32 --steps 5 --warmup 2 > gpurun_out/rehearse/rccl1_init.log 2>&1 || { tail -15 gpurun_out/rehearse/rccl1_init.log; exit 1; }
echo "rccl1_init: $(grep -c 'Init COMPLETE\|ncclCommInitRank.*COMPLETE\|Init START' gpurun_out/rehearse/rccl1_init.log) RCCL init lines; $(grep -o '"comm_layer": "[^"]*"' gpurun_out/rehearse/rccl1_init.log | tail -1)"
grep -i 'init complete\|init start\|commInitRank' gpurun_out/rehearse/rccl1_init.log | cut -c1-160 | head -6
# the fallback: the native self-test fails on purpose, every rank adopts a c10d RCCL group instead
DPA_FORCE_COLLECTIVES=1 DPA_COMM_SELFTEST_TIMEOUT=0 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --model tiny_resnet --batch 32 --steps 5 --warmup 2 > gpurun_out/rehearse/rccl1_fallback.log 2>&1 || { tail -15 gpurun_out/rehearse/rccl1_fallback.log; exit 1; }
echo "rccl1_fallback: $(grep -o '"comm_layer": "[^"]*"' gpurun_out/rehearse/rccl1_fallback.log | tail -1) $(grep -c 'falling back to c10d' gpurun_out/rehearse/rccl1_fallback.log) fallback notice(s)"
# GPT-2 over the real RCCL call path at world size 1: the weight-gradient batcher (ops/wgroup.py) feeding buckets that are
# all-reduced for real, optimizer overlapped with backward
DPA_FORCE_COLLECTIVES=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --model gpt2 --steps 10 --warmup 3 > gpurun_out/rehearse/rccl1_gpt2.log 2>&1 || { tail -15 gpurun_out/rehearse/rccl1_gpt2.log; exit 1; }
echo "rccl1_gpt2 $(grep '^{' gpurun_out/rehearse/rccl1_gpt2.log | tail -1 | cut -c1-150)"
grep -o '"allreduce_calls_per_step": [0-9.]*' gpurun_out/rehearse/rccl1_gpt2.log | tail -1
