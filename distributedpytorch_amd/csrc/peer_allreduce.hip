// One-kernel all-reduce over peer-mapped memory.
//
// Every rank owns a STAGING buffer and a small SIGNAL block in its own HBM; both are exported with hipIpcGetMemHandle
// and mapped by every other rank of the group (csrc/peer_comm.cpp), so a kernel on GPU r dereferences pointers into the
// memory of GPUs 0..W-1 directly: on the MI355X mesh each of those pointers is one xGMI link.  One launch per rank does
// the whole two-shot all-reduce (sum) of an fp32 tensor, in place:
//
//   phase 0  my values of every OTHER rank's shard -> my staging buffer (local copy; fp32, or bf16 on the wire);
//   barrier A
//   phase 1  shard `rank`: my own fp32 values + the W-1 peers' staged copies, read over the W-1 links AT ONCE, summed in
//            fp32, scaled, written to the tensor (final) and to my staging buffer (for the peers);
//   barrier B
//   phase 2  every other shard p: the reduced values from rank p's staging buffer -> the tensor.
//
// Wire bytes per rank: 2 * (W-1)/W * S, all links busy in both phases - the direct algorithm of rccl_comm.cpp's
// all_reduce_direct without its two grouped RCCL send/recv rounds, its three local kernels and its scratch traffic.
//
// Slicing: block b of EVERY rank handles the same slice [u0, u1) of every shard, in all three phases.  What block b
// of rank r reads from rank p's staging buffer was therefore written by block b of rank p, so the barriers are
// pairwise per block index: no grid-wide synchronisation on any GPU, and a block that is scheduled late (the compute
// stream owns most CUs during backward) only delays its own slice.
//
// Barrier = the hand-off of cdna_hip_programming.md Guideline 16 at SYSTEM scope: every storing wave has drained
// (__syncthreads), one lane releases (buffer_wbl2: my dirty L2 lines reach HBM, where the links read), then lanes
// 0..W-1 each raise my slot in one peer's signal block with a system-scope store and poll that peer's slot in mine
// (signal memory is uncached fine-grained memory: hipDeviceMallocUncached), then one acquire (buffer_inv) before any
// lane loads.  Slots hold the launch sequence number, which only grows: no reset pass, no ABA.  Sequence numbers start at
// a random EPOCH BASE drawn per communicator group (peer_comm.cpp set_base) and a poll accepts a slot only inside a window
// of 2^20 launches above the number it waits for: the protocol does not depend on what the signal memory held before.
// (It did, once: a second job on the same GPU found slot values of the job before it - larger than its own first sequence
// numbers - although the block is zeroed at allocation; its polls passed at once, one rank copied a shard that had not
// been reduced yet and the other waited for a flag that had been "seen" already: profiles/r3/peer_shared_gpu_flake.txt.)
// The staging buffer has two halves used by alternate launches: rank p may still be reading my half of launch k in
// its phase 2 when I enter launch k + 1 (which fills the other half); it cannot be later than that, because my
// barrier B of launch k + 1 needs its flag, which it raises after ITS phase 1 of k + 1.
//
// Every poll is bounded (s_memrealtime, 100 MHz): a peer that never arrives sets the error word of the signal block
// and the kernel runs to completion with wrong data instead of hanging the queue - the host checks the word after the
// start-up self-test and at synchronisation points (Communicator.attach_peer falls back to RCCL on any error).
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int SIG_SLOTS = 16;                                 // ranks per block row of the signal arrays

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// which: 0 = barrier A, 1 = barrier B.  Wave 0 does the signalling; called by every thread of the block.
__device__ __forceinline__ void peer_barrier(const PeerArgs& a, int which) {
    const int tid = threadIdx.x;
    const unsigned off = (unsigned)((which * PEER_MAX_BLOCKS + (int)blockIdx.x) * SIG_SLOTS);
    __syncthreads();                                           // every wave's stores have been issued and waited for
    if (tid < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: write back what the links will read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler may drop the fence's own wait)
        if (tid < a.world - 1) {                               // lane i <-> peer i of the rank-ordered peer list
            const int p = tid < a.rank ? tid : tid + 1;
            st_sys(a.sig_peer[tid] + off + a.rank, a.seq);
            const unsigned* mine = a.sig_self + off + p;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while ((unsigned)(ld_sys(mine) - a.seq) >= PEER_SEQ_WINDOW) {       // not this epoch's number, or not yet raised
                if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                    st_sys(a.sig_self + PEER_ERR_WORD, a.seq);  // host-visible: rank p never arrived at this barrier
                    st_sys(a.sig_self + PEER_ERR_WORD + 1, (unsigned)((which << 16) | ((int)blockIdx.x << 8) | p));
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope: drop every cached copy of peer data
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int VE>
__device__ __forceinline__ void load_user(const float* user, long e, long n, float (&f)[VE]) {
    if (e + VE <= n) {
#pragma unroll
        for (int q = 0; q < VE / 4; ++q) {
            const float4 v = *(const float4*)(user + e + 4 * q);
            f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) f[j] = e + j < n ? user[e + j] : 0.f;
    }
}

template <int VE>
__device__ __forceinline__ void store_user(float* user, long e, long n, const float (&f)[VE]) {
    if (e + VE <= n) {
#pragma unroll
        for (int q = 0; q < VE / 4; ++q) *(float4*)(user + e + 4 * q) = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) if (e + j < n) user[e + j] = f[j];
    }
}

// one 16-byte staging unit <-> VE floats
template <bool BF16>
__device__ __forceinline__ void unit_to_floats(const uint4& v, float (&f)[BF16 ? 8 : 4]) {
    if constexpr (BF16) unpack8(v, f);
    else { f[0] = __builtin_bit_cast(float, v.x); f[1] = __builtin_bit_cast(float, v.y); f[2] = __builtin_bit_cast(float, v.z); f[3] = __builtin_bit_cast(float, v.w); }
}
template <bool BF16>
__device__ __forceinline__ uint4 floats_to_unit(const float (&f)[BF16 ? 8 : 4]) {
    if constexpr (BF16) return pack8(f);
    else return make_uint4(__builtin_bit_cast(unsigned, f[0]), __builtin_bit_cast(unsigned, f[1]), __builtin_bit_cast(unsigned, f[2]), __builtin_bit_cast(unsigned, f[3]));
}

template <bool BF16>
__global__ __launch_bounds__(512) void peer_allreduce_kernel(PeerArgs a) {
    constexpr int VE = BF16 ? 8 : 4;                           // tensor elements per 16-byte staging unit
    const int tid = threadIdx.x, nb = (int)gridDim.x, W = a.world, me = a.rank;
    const long units = a.shard / VE;                           // units per shard (shard % 8 == 0: host)
    const long per = (units + nb - 1) / nb;
    const long u0 = min(units, (long)blockIdx.x * per), u1 = min(units, u0 + per);
    uint4* const mine = (uint4*)a.stage_self;

    // ---- phase 0: my contribution to every other rank's shard ----
    for (int s = 0; s < W; ++s) {
        if (s == me) continue;
        const long ub = (long)s * units;
        for (long u = u0 + tid; u < u1; u += 512) {
            float f[VE];
            load_user<VE>(a.user, (ub + u) * VE, a.n, f);
            mine[ub + u] = floats_to_unit<BF16>(f);
        }
    }
    peer_barrier(a, 0);

    // ---- phase 1: my shard = my own values + every peer's staged copy, W - 1 links read at once ----
    {
        const long ub = (long)me * units;
        for (long u = u0 + tid; u < u1; u += 512) {
            float acc[VE];
            load_user<VE>(a.user, (ub + u) * VE, a.n, acc);
            uint4 v[PEER_MAX_RANKS - 1];
#pragma unroll
            for (int i = 0; i < PEER_MAX_RANKS - 1; ++i) {
                if (i < W - 1) v[i] = ((const uint4*)a.stage_peer[i])[ub + u];      // peers in rank order
            }
#pragma unroll
            for (int i = 0; i < PEER_MAX_RANKS - 1; ++i) {
                if (i < W - 1) {
                    float f[VE];
                    unit_to_floats<BF16>(v[i], f);
#pragma unroll
                    for (int j = 0; j < VE; ++j) acc[j] += f[j];
                }
            }
#pragma unroll
            for (int j = 0; j < VE; ++j) acc[j] *= a.scale;
            const uint4 r = floats_to_unit<BF16>(acc);
            mine[ub + u] = r;
            if constexpr (BF16) unit_to_floats<BF16>(r, acc);  // the owner keeps the ROUNDED value too: replicas stay bit-identical
            store_user<VE>(a.user, (ub + u) * VE, a.n, acc);
        }
    }
    peer_barrier(a, 1);

    // ---- phase 2: the other shards, reduced, from their owners ----
    for (long u = u0 + tid; u < u1; u += 512) {
        uint4 v[PEER_MAX_RANKS - 1];
#pragma unroll
        for (int i = 0; i < PEER_MAX_RANKS - 1; ++i) {
            const int p = i < me ? i : i + 1;
            if (i < W - 1) v[i] = ((const uint4*)a.stage_peer[i])[(long)p * units + u];
        }
#pragma unroll
        for (int i = 0; i < PEER_MAX_RANKS - 1; ++i) {
            const int p = i < me ? i : i + 1;
            if (i < W - 1) {
                float f[VE];
                unit_to_floats<BF16>(v[i], f);
                store_user<VE>(a.user, ((long)p * units + u) * VE, a.n, f);
            }
        }
    }
}

}  // namespace

// Blocks of a launch: a pure function of the element count, so every rank of a group picks the same grid (the slicing
// and the per-block signal slots depend on it).
int peer_allreduce_blocks(long n, int max_blocks) {
    if (max_blocks > PEER_MAX_BLOCKS) max_blocks = PEER_MAX_BLOCKS;
    if (max_blocks < 1) max_blocks = 1;
    long nb = (n * 4 + (256 << 10) - 1) / (256 << 10);         // one block per 256 KiB of tensor
    if (nb < 1) nb = 1;
    if (nb > max_blocks) nb = max_blocks;
    return (int)nb;
}

void peer_allreduce(const PeerArgs& a, int blocks, int bf16_wire, hipStream_t st) {
    if (bf16_wire) hipLaunchKernelGGL(peer_allreduce_kernel<true>, dim3(blocks), dim3(512), 0, st, a);
    else           hipLaunchKernelGGL(peer_allreduce_kernel<false>, dim3(blocks), dim3(512), 0, st, a);
}

}  // namespace dpa
