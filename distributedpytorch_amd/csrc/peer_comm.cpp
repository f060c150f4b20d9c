// Peer-mapped all-reduce: host side (kernel and protocol: peer_allreduce.hip).
//
//   * every rank allocates a staging buffer (two halves of `chunk` bytes, plain device memory) and a signal block
//     (uncached fine-grained device memory: flags other GPUs raise must be seen by a running kernel), exports both with
//     hipIpcGetMemHandle; the 64-byte handles travel over the control plane (comm.py: the gloo group or the TCP store)
//     and every rank maps every peer's two allocations with hipIpcOpenMemHandle - on the 8-GPU mesh that is one xGMI
//     link per mapping;
//   * all_reduce(tensor) enqueues ONE kernel per <= chunk bytes of the tensor on this object's communication stream,
//     ordered behind the producer by an event and handed back as a ticket the consumer's stream waits on - the same
//     stream / event / caching-allocator discipline as rccl_comm.cpp;
//   * no RCCL call anywhere: the object works without a RCCL communicator (all ranks of a test on ONE GPU, where RCCL
//     refuses to initialise), and beside one (DPA_ALLREDUCE=ipc: the native communicator keeps every other collective).
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

namespace {

using at::Tensor;
using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;

#define PEER_HIP_CHECK(cmd) do { hipError_t e_ = (cmd); TORCH_CHECK(e_ == hipSuccess, "HIP error: ", hipGetErrorString(e_), " at " #cmd); } while (0)

class PeerAllReduce {
  public:
    PeerAllReduce(int rank, int world, int device, int64_t chunk_bytes, int max_blocks, double timeout_s)
        : rank_(rank), world_(world), device_(device), max_blocks_(max_blocks),
          stream_(c10::hip::getStreamFromPoolMasqueradingAsCUDA(/*isHighPriority=*/true, (c10::DeviceIndex)device)) {
        TORCH_CHECK(world >= 1 && world <= dpa::PEER_MAX_RANKS, "PeerAllReduce: 1..", dpa::PEER_MAX_RANKS, " ranks (one xGMI node), got ", world);
        TORCH_CHECK(rank >= 0 && rank < world, "PeerAllReduce: bad rank");
        TORCH_CHECK(chunk_bytes >= (1 << 16), "PeerAllReduce: chunk of at least 64 KiB");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        // one chunk of tensor = world shards of ceil(n / world) elements rounded up to 8: 32 bytes of padding per rank
        chunk_elems_ = chunk_bytes / 4;
        half_bytes_ = (size_t)chunk_bytes + (size_t)world * 64 + 256;
        half_bytes_ = (half_bytes_ + 255) / 256 * 256;
        PEER_HIP_CHECK(hipMalloc(&stage_, 2 * half_bytes_));
        hipError_t e = hipExtMallocWithFlags((void**)&signal_, dpa::PEER_SIGNAL_WORDS * sizeof(unsigned), hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipFree(stage_);
            TORCH_CHECK(false, "PeerAllReduce: no uncached device memory for the signal block: ", hipGetErrorString(e));
        }
        PEER_HIP_CHECK(hipMemset(stage_, 0, 2 * half_bytes_));
        PEER_HIP_CHECK(hipMemset(signal_, 0, dpa::PEER_SIGNAL_WORDS * sizeof(unsigned)));
        PEER_HIP_CHECK(hipDeviceSynchronize());
        timeout_ticks_ = (unsigned long long)(timeout_s * 1e8);
        // First launch of the kernel on this stream, alone (world 1: no flag, no poll): code object load and hardware-queue
        // creation happen HERE, not in front of a launch whose peers are already polling for this rank.  (Measured with four
        // ranks sharing one GPU: a rank's first launch call blocked for as long as the other ranks' kernels kept spinning.)
        {
            dpa::PeerArgs a{};
            a.rank = 0; a.world = 1; a.seq = 0; a.scale = 1.f;
            a.n = 64; a.shard = 64;
            a.user = (float*)stage_;
            a.stage_self = (char*)stage_ + half_bytes_;
            a.sig_self = signal_;
            a.timeout_ticks = timeout_ticks_;
            dpa::peer_allreduce(a, 1, 0, stream_.stream());
            PEER_HIP_CHECK(hipGetLastError());
            PEER_HIP_CHECK(hipStreamSynchronize(stream_.stream()));
        }
        peer_stage_.assign(world, nullptr);
        peer_signal_.assign(world, nullptr);
        peer_stage_[rank] = stage_;
        peer_signal_[rank] = signal_;
    }

    ~PeerAllReduce() {
        (void)hipStreamSynchronize(stream_.stream());
        for (int p = 0; p < world_; ++p) {
            if (p == rank_) continue;
            if (peer_stage_[p] != nullptr) (void)hipIpcCloseMemHandle(peer_stage_[p]);
            if (peer_signal_[p] != nullptr) (void)hipIpcCloseMemHandle(peer_signal_[p]);
        }
        if (stage_ != nullptr) (void)hipFree(stage_);
        if (signal_ != nullptr) (void)hipFree(signal_);
    }

    // (staging handle, signal handle) of this rank, 64 bytes each
    std::pair<py::bytes, py::bytes> handles() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        hipIpcMemHandle_t hs, hg;
        PEER_HIP_CHECK(hipIpcGetMemHandle(&hs, stage_));
        PEER_HIP_CHECK(hipIpcGetMemHandle(&hg, signal_));
        return {py::bytes(reinterpret_cast<const char*>(&hs), sizeof(hs)), py::bytes(reinterpret_cast<const char*>(&hg), sizeof(hg))};
    }

    // handles of every rank of the group, indexed by rank (this rank's own entry is ignored)
    void open(const std::vector<std::pair<std::string, std::string>>& all) {
        TORCH_CHECK((int)all.size() == world_, "PeerAllReduce.open: one handle pair per rank");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        for (int p = 0; p < world_; ++p) {
            if (p == rank_) continue;
            TORCH_CHECK(all[p].first.size() == sizeof(hipIpcMemHandle_t) && all[p].second.size() == sizeof(hipIpcMemHandle_t),
                        "PeerAllReduce.open: handles are ", sizeof(hipIpcMemHandle_t), " bytes");
            hipIpcMemHandle_t hs, hg;
            std::memcpy(&hs, all[p].first.data(), sizeof(hs));
            std::memcpy(&hg, all[p].second.data(), sizeof(hg));
            PEER_HIP_CHECK(hipIpcOpenMemHandle(&peer_stage_[p], hs, hipIpcMemLazyEnablePeerAccess));
            PEER_HIP_CHECK(hipIpcOpenMemHandle((void**)&peer_signal_[p], hg, hipIpcMemLazyEnablePeerAccess));
        }
        opened_ = true;
    }

    // In-place sum of an fp32 tensor over the group, times `scale`; bf16_wire: bf16 on the links, fp32 accumulation.
    // Every rank of the group must make the same sequence of calls with the same sizes (a collective).
    int64_t all_reduce(Tensor t, double scale, bool bf16_wire) {
        TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat, "PeerAllReduce: contiguous fp32 device tensor");
        TORCH_CHECK(t.device().index() == device_, "PeerAllReduce: tensor on another device");
        TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, "PeerAllReduce: tensor must be 16-byte aligned");
        TORCH_CHECK(opened_ || world_ == 1, "PeerAllReduce: open() the peers' handles first");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        Stream cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_);
        ready_.record(cur);
        ready_.block(stream_);
        if (world_ > 1) {
            float* base = t.data_ptr<float>();
            const int64_t n = t.numel();
            // bf16 staging holds twice the elements per byte, but keep ONE chunking for both wire formats
            for (int64_t lo = 0; lo < n; lo += chunk_elems_) {
                const int64_t len = std::min<int64_t>(chunk_elems_, n - lo);
                dpa::PeerArgs a{};
                a.rank = rank_; a.world = world_;
                a.seq = ++seq_;
                a.scale = (float)scale;
                a.n = len;
                a.shard = ((len + world_ - 1) / world_ + 7) / 8 * 8;
                a.user = base + lo;
                const size_t half = (size_t)(a.seq & 1u) * half_bytes_;
                a.stage_self = (char*)stage_ + half;
                a.sig_self = signal_;
                int i = 0;
                for (int p = 0; p < world_; ++p) {
                    if (p == rank_) continue;
                    a.stage_peer[i] = (char*)peer_stage_[p] + half;
                    a.sig_peer[i] = peer_signal_[p];
                    ++i;
                }
                a.timeout_ticks = timeout_ticks_;
                dpa::peer_allreduce(a, dpa::peer_allreduce_blocks(len, max_blocks_), bf16_wire ? 1 : 0, stream_.stream());
                PEER_HIP_CHECK(hipGetLastError());
            }
        } else if (scale != 1.0) {
            dpa::scale_f32(t.data_ptr<float>(), t.numel(), (float)scale, stream_.stream());
        }
        c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(t.storage().data_ptr(), stream_);
        for (auto it = pending_.begin(); it != pending_.end();)
            it = it->second->query() ? pending_.erase(it) : std::next(it);
        auto done = std::make_unique<at::cuda::CUDAEvent>(hipEventDisableTiming);
        done->record(stream_);
        const int64_t ticket = next_ticket_++;
        pending_[ticket] = std::move(done);
        return ticket;
    }

    void wait(int64_t ticket) {
        auto it = pending_.find(ticket);
        if (it == pending_.end()) return;
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        it->second->block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_));
    }

    bool query(int64_t ticket) {
        auto it = pending_.find(ticket);
        return it == pending_.end() ? true : it->second->query();
    }

    void synchronize() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        PEER_HIP_CHECK(hipStreamSynchronize(stream_.stream()));
        pending_.clear();
    }

    // Sequence numbers of this group start at `base` (the same random number on every rank, agreed over the control
    // plane before the first collective): slot values of an EARLIER job that may survive in the signal memory are outside
    // the acceptance window of this one (peer_allreduce.hip).  Call once, before the first all_reduce.
    void set_base(int64_t base) {
        TORCH_CHECK(seq_ == 0 && base_ == 0, "PeerAllReduce.set_base: once, before the first collective");
        TORCH_CHECK(base >= (int64_t)dpa::PEER_SEQ_WINDOW && base < (int64_t)0x7fffffff, "PeerAllReduce.set_base: 2^20 <= base < 2^31");
        base_ = seq_ = (unsigned)base;
    }

    // launch number (counted from 1) at which a bounded poll of this rank last ran out (0 = never): a peer was missing
    // from a barrier.  Only a word written in THIS epoch counts.
    int64_t error_seq() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        unsigned v = 0;
        PEER_HIP_CHECK(hipMemcpy(&v, signal_ + dpa::PEER_ERR_WORD, sizeof(v), hipMemcpyDeviceToHost));
        const unsigned d = v - base_;
        return (v != 0 && d >= 1 && d < dpa::PEER_SEQ_WINDOW) ? (int64_t)d : 0;
    }

    // (barrier << 16 | block << 8 | missing rank) of that poll
    int64_t error_detail() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        unsigned v = 0;
        PEER_HIP_CHECK(hipMemcpy(&v, signal_ + dpa::PEER_ERR_WORD + 1, sizeof(v), hipMemcpyDeviceToHost));
        return (int64_t)v;
    }

    // diagnostics (scripts/peer_probe.py): one 32-bit word of rank p's staging buffer / signal block through THIS rank's mapping
    int64_t debug_read(int p, bool signal, int64_t word) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        TORCH_CHECK(p >= 0 && p < world_, "bad rank");
        const unsigned* src = (signal ? peer_signal_[p] : (const unsigned*)peer_stage_[p]) + word;
        unsigned v = 0;
        PEER_HIP_CHECK(hipMemcpy(&v, src, sizeof(v), hipMemcpyDeviceToHost));
        return (int64_t)v;
    }
    void debug_write(int p, bool signal, int64_t word, int64_t value) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        TORCH_CHECK(p >= 0 && p < world_, "bad rank");
        unsigned* dst = (signal ? peer_signal_[p] : (unsigned*)peer_stage_[p]) + word;
        const unsigned v = (unsigned)value;
        PEER_HIP_CHECK(hipMemcpy(dst, &v, sizeof(v), hipMemcpyHostToDevice));
        PEER_HIP_CHECK(hipDeviceSynchronize());
    }

    int64_t launches() const { return (int64_t)(seq_ - base_); }
    int rank() const { return rank_; }
    int world_size() const { return world_; }
    int64_t chunk_elems() const { return chunk_elems_; }

  private:
    c10::Device dev() const { return c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device_); }

    int rank_, world_, device_, max_blocks_;
    Stream stream_;
    void* stage_ = nullptr;
    unsigned* signal_ = nullptr;
    size_t half_bytes_ = 0;
    int64_t chunk_elems_ = 0;
    unsigned long long timeout_ticks_ = 0;
    std::vector<void*> peer_stage_;
    std::vector<unsigned*> peer_signal_;
    bool opened_ = false;
    unsigned seq_ = 0, base_ = 0;
    at::cuda::CUDAEvent ready_{hipEventDisableTiming};
    std::map<int64_t, std::unique_ptr<at::cuda::CUDAEvent>> pending_;
    int64_t next_ticket_ = 1;
};

}  // namespace

void bind_peer(pybind11::module_& m) {
    m.attr("PEER_MAX_RANKS") = (int)dpa::PEER_MAX_RANKS;
    py::class_<PeerAllReduce>(m, "PeerAllReduce")
        .def(py::init<int, int, int, int64_t, int, double>(), py::arg("rank"), py::arg("world_size"), py::arg("device"),
             py::arg("chunk_bytes") = (int64_t)64 << 20, py::arg("max_blocks") = 32, py::arg("timeout_s") = 30.0)
        .def("handles", &PeerAllReduce::handles)
        .def("open", &PeerAllReduce::open)
        .def("all_reduce", &PeerAllReduce::all_reduce, py::arg("tensor"), py::arg("scale") = 1.0, py::arg("bf16_wire") = false)
        .def("wait", &PeerAllReduce::wait)
        .def("query", &PeerAllReduce::query)
        .def("synchronize", &PeerAllReduce::synchronize)
        .def("set_base", &PeerAllReduce::set_base)
        .def("error_seq", &PeerAllReduce::error_seq)
        .def("error_detail", &PeerAllReduce::error_detail)
        .def("debug_read", &PeerAllReduce::debug_read)
        .def("debug_write", &PeerAllReduce::debug_write)
        .def_property_readonly("launches", &PeerAllReduce::launches)
        .def_property_readonly("rank", &PeerAllReduce::rank)
        .def_property_readonly("world_size", &PeerAllReduce::world_size)
        .def_property_readonly("chunk_elems", &PeerAllReduce::chunk_elems);
}
