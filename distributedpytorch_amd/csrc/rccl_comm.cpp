// Native RCCL communicator (optional backend, DPA_COMM=native).
//
// torch.distributed's "nccl" backend is RCCL already and is the default path of this framework; this class
// is the thin C++ layer SURVEY.md §5.8 asks for when the collective ALGORITHM has to be ours:
//   * one RCCL communicator per process (bootstrapped from a 128-byte unique id that travels through the
//     existing rendezvous), one dedicated high-priority HIP stream, hipEvent hand-offs in both directions
//     (compute -> comm when a bucket is ready, comm -> compute when it is reduced): no host blocking;
//   * ring collectives straight from RCCL (all_reduce / broadcast / reduce_scatter / all_gather, send / recv,
//     grouped P2P);
//   * all_reduce_direct: the one-shot algorithm for the fully connected xGMI mesh (7 point-to-point links per
//     MI355X): every rank pushes shard j to rank j over the direct link (grouped send/recv = all links busy
//     at once), reduces its own shard locally in fp32, and pushes the result back the same way.  Wire time
//     ~ 2 * (S / N) per link instead of 2 * (N - 1) / N * S over one ring.
// Links against the RCCL that torch itself loaded (torch/lib/librccl.so), so there is one RCCL in the process.
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace {

using at::Tensor;

#define RCCL_CHECK(cmd) do { ncclResult_t r_ = (cmd); TORCH_CHECK(r_ == ncclSuccess, "RCCL error: ", ncclGetErrorString(r_), " at " #cmd); } while (0)
#define HIP_CHECK(cmd) do { hipError_t e_ = (cmd); TORCH_CHECK(e_ == hipSuccess, "HIP error: ", hipGetErrorString(e_), " at " #cmd); } while (0)

ncclDataType_t nccl_dtype(const Tensor& t) {
    switch (t.scalar_type()) {
        case at::kFloat: return ncclFloat;
        case at::kBFloat16: return ncclBfloat16;
        case at::kHalf: return ncclHalf;
        case at::kDouble: return ncclDouble;
        case at::kLong: return ncclInt64;
        case at::kInt: return ncclInt32;
        case at::kByte: return ncclUint8;
        case at::kChar: return ncclInt8;
        default: TORCH_CHECK(false, "unsupported dtype for RCCL: ", t.scalar_type());
    }
}

class RcclComm {
  public:
    static py::bytes unique_id() {
        ncclUniqueId id;
        RCCL_CHECK(ncclGetUniqueId(&id));
        return py::bytes(reinterpret_cast<const char*>(&id), sizeof(id));
    }

    RcclComm(const std::string& id_bytes, int rank, int world, int device) : rank_(rank), world_(world), device_(device) {
        TORCH_CHECK(id_bytes.size() == sizeof(ncclUniqueId), "unique id must be ", sizeof(ncclUniqueId), " bytes");
        TORCH_CHECK(rank >= 0 && rank < world, "bad rank");
        ncclUniqueId id;
        std::memcpy(&id, id_bytes.data(), sizeof(id));
        HIP_CHECK(hipSetDevice(device));
        int lo = 0, hi = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));     // hi = numerically lowest = highest priority
        const char* pr = std::getenv("DPA_COMM_PRIO");          // "high" (default) | "normal"
        const bool normal = pr != nullptr && std::string(pr) == "normal";
        HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, normal ? lo : hi));
        RCCL_CHECK(ncclCommInitRank(&comm_, world, id, rank));
    }

    ~RcclComm() {
        if (comm_ != nullptr) {
            (void)hipStreamSynchronize(stream_);
            (void)ncclCommDestroy(comm_);
        }
        for (auto& kv : pending_) (void)hipEventDestroy(kv.second.done);
        for (hipEvent_t e : retired_) (void)hipEventDestroy(e);
        if (stream_ != nullptr) (void)hipStreamDestroy(stream_);
    }

    int rank() const { return rank_; }
    int world_size() const { return world_; }

    int64_t all_reduce(Tensor t, int op) {
        check(t);
        begin();
        RCCL_CHECK(ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), nccl_dtype(t), red_op(op), comm_, stream_));
        return finish({t});
    }

    int64_t broadcast(Tensor t, int root) {
        check(t);
        begin();
        RCCL_CHECK(ncclBroadcast(t.data_ptr(), t.data_ptr(), t.numel(), nccl_dtype(t), root, comm_, stream_));
        return finish({t});
    }

    int64_t reduce_scatter(Tensor out, Tensor in) {
        check(out); check(in);
        TORCH_CHECK(in.numel() == out.numel() * world_ && in.scalar_type() == out.scalar_type(), "reduce_scatter: sizes");
        begin();
        RCCL_CHECK(ncclReduceScatter(in.data_ptr(), out.data_ptr(), out.numel(), nccl_dtype(out), ncclSum, comm_, stream_));
        return finish({out, in});
    }

    int64_t all_gather(Tensor out, Tensor in) {
        check(out); check(in);
        TORCH_CHECK(out.numel() == in.numel() * world_ && in.scalar_type() == out.scalar_type(), "all_gather: sizes");
        begin();
        RCCL_CHECK(ncclAllGather(in.data_ptr(), out.data_ptr(), in.numel(), nccl_dtype(in), comm_, stream_));
        return finish({out, in});
    }

    // ops: list of (is_send, tensor, peer); issued as ONE RCCL group (pipeline stage boundaries)
    int64_t batch_p2p(const std::vector<std::tuple<bool, Tensor, int>>& ops) {
        std::vector<Tensor> keep;
        for (auto& o : ops) { check(std::get<1>(o)); keep.push_back(std::get<1>(o)); }
        begin();
        RCCL_CHECK(ncclGroupStart());
        for (auto& o : ops) {
            const Tensor& t = std::get<1>(o);
            const int peer = std::get<2>(o);
            TORCH_CHECK(peer >= 0 && peer < world_ && peer != rank_, "batch_p2p: bad peer ", peer);
            if (std::get<0>(o)) RCCL_CHECK(ncclSend(t.data_ptr(), t.numel(), nccl_dtype(t), peer, comm_, stream_));
            else                RCCL_CHECK(ncclRecv(t.data_ptr(), t.numel(), nccl_dtype(t), peer, comm_, stream_));
        }
        RCCL_CHECK(ncclGroupEnd());
        return finish(keep);
    }

    // One-shot all-reduce (sum) of an fp32 buffer for a fully connected topology.  `scratch` holds the
    // (world - 1) incoming copies of this rank's shard: numel >= (world - 1) * ceil(n / world).
    int64_t all_reduce_direct(Tensor t, Tensor scratch) {
        check(t); check(scratch);
        TORCH_CHECK(t.scalar_type() == at::kFloat && scratch.scalar_type() == at::kFloat, "all_reduce_direct: fp32 buffers");
        const int64_t n = t.numel();
        const int64_t s = (n + world_ - 1) / world_;
        TORCH_CHECK(scratch.numel() >= (int64_t)(world_ - 1) * s, "all_reduce_direct: scratch too small");
        auto lo = [&](int r) { return std::min<int64_t>(n, (int64_t)r * s); };
        auto len = [&](int r) { return std::min<int64_t>(n, (int64_t)(r + 1) * s) - lo(r); };
        float* base = t.data_ptr<float>();
        float* scr = scratch.data_ptr<float>();
        begin();
        if (world_ > 1) {
            const int64_t mine = len(rank_);
            // phase 1: shard j -> rank j over the direct link; everyone's copy of MY shard arrives in scratch
            RCCL_CHECK(ncclGroupStart());
            int slot = 0;
            for (int p = 0; p < world_; ++p) {
                if (p == rank_) continue;
                if (len(p) > 0) RCCL_CHECK(ncclSend(base + lo(p), len(p), ncclFloat, p, comm_, stream_));
                if (mine > 0) RCCL_CHECK(ncclRecv(scr + (int64_t)slot * s, mine, ncclFloat, p, comm_, stream_));
                ++slot;
            }
            RCCL_CHECK(ncclGroupEnd());
            // local N-way reduction in fp32 on the comm stream
            if (mine > 0) dpa::reduce_shards(base + lo(rank_), scr, world_ - 1, s, mine, stream_);
            // phase 2: the reduced shard goes back to everyone, theirs arrive in place
            RCCL_CHECK(ncclGroupStart());
            for (int p = 0; p < world_; ++p) {
                if (p == rank_) continue;
                if (mine > 0) RCCL_CHECK(ncclSend(base + lo(rank_), mine, ncclFloat, p, comm_, stream_));
                if (len(p) > 0) RCCL_CHECK(ncclRecv(base + lo(p), len(p), ncclFloat, p, comm_, stream_));
            }
            RCCL_CHECK(ncclGroupEnd());
        }
        return finish({t, scratch});
    }

    // the CURRENT compute stream waits for the operation behind `ticket` (no host block)
    void wait(int64_t ticket) {
        auto it = pending_.find(ticket);
        if (it == pending_.end()) return;                     // already waited for
        c10::hip::HIPGuardMasqueradingAsCUDA guard(c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device_));
        hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
        HIP_CHECK(hipStreamWaitEvent(cur, it->second.done, 0));
        retire(it->second.done);
        pending_.erase(it);
    }

    void synchronize() {
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (auto& kv : pending_) HIP_CHECK(hipEventDestroy(kv.second.done));
        pending_.clear();
    }

  private:
    struct Pending { hipEvent_t done; std::vector<Tensor> keep; };

    void check(const Tensor& t) const {
        TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "RcclComm: contiguous device tensors only");
        TORCH_CHECK(t.device().index() == device_, "RcclComm: tensor on device ", (int)t.device().index(), ", communicator on ", device_);
    }
    static ncclRedOp_t red_op(int op) { return op == 1 ? ncclMax : op == 2 ? ncclMin : ncclSum; }

    // Events are never re-recorded: a stream-wait on ROCm 7 followed the event's LATEST record when it finally executed,
    // not the record that was current when the wait was enqueued, so re-recording a pooled event under a queued wait
    // tied that wait to work submitted later - with bucket-sized (32 MB) operands the queues deadlocked.  Each use gets
    // a fresh event, destroyed right after its wait is enqueued (destruction of an event in use is deferred by the
    // runtime); creation costs about a microsecond, ten times per step.
    // A retired event still has a wait queued on some stream.  It is destroyed only after BOTH streams have drained:
    // every kRetireBatch events (about 200 ResNet-50 steps) the communicator synchronises its stream and the current
    // compute stream once - a few milliseconds every few seconds - and frees the batch.
    static constexpr size_t kRetireBatch = 2048;
    void retire(hipEvent_t e) {
        retired_.push_back(e);
        if (retired_.size() >= kRetireBatch) collect();
    }
    void collect() {
        hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
        HIP_CHECK(hipStreamSynchronize(stream_));
        HIP_CHECK(hipStreamSynchronize(cur));
        for (hipEvent_t e : retired_) (void)hipEventDestroy(e);
        retired_.clear();
    }

    static hipEvent_t new_event() {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    }

    // comm stream waits for everything enqueued on the compute stream so far (the producer of the buffer)
    void begin() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device_));
        hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
        hipEvent_t ready = new_event();
        HIP_CHECK(hipEventRecord(ready, cur));
        HIP_CHECK(hipStreamWaitEvent(stream_, ready, 0));
        retire(ready);
    }

    int64_t finish(std::vector<Tensor> keep) {
        hipEvent_t done = new_event();
        HIP_CHECK(hipEventRecord(done, stream_));
        const int64_t ticket = next_ticket_++;
        pending_[ticket] = Pending{done, std::move(keep)};
        return ticket;
    }


    int rank_, world_, device_;
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    std::map<int64_t, Pending> pending_;
    std::vector<hipEvent_t> retired_;
    int64_t next_ticket_ = 1;
};

}  // namespace

void bind_rccl(pybind11::module_& m) {
    py::class_<RcclComm>(m, "RcclComm")
        .def(py::init<const std::string&, int, int, int>(), py::arg("unique_id"), py::arg("rank"), py::arg("world_size"), py::arg("device"))
        .def_static("unique_id", &RcclComm::unique_id)
        .def_property_readonly("rank", &RcclComm::rank)
        .def_property_readonly("world_size", &RcclComm::world_size)
        .def("all_reduce", &RcclComm::all_reduce, py::arg("tensor"), py::arg("op") = 0)
        .def("all_reduce_direct", &RcclComm::all_reduce_direct)
        .def("broadcast", &RcclComm::broadcast)
        .def("reduce_scatter", &RcclComm::reduce_scatter)
        .def("all_gather", &RcclComm::all_gather)
        .def("batch_p2p", &RcclComm::batch_p2p)
        .def("wait", &RcclComm::wait)
        .def("synchronize", &RcclComm::synchronize);
}
