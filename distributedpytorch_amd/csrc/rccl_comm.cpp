// Native RCCL communicator: the framework's own C++ layer over RCCL.
//
//   * one RCCL communicator per process group (bootstrapped from a 128-byte unique id that travels through
//     the rendezvous store, distributedpytorch_amd/store.py, or through an existing process group), one
//     dedicated high-priority HIP stream, event hand-offs in both directions (compute -> comm when a bucket
//     is ready, comm -> compute when it is reduced): no host blocking anywhere in a step;
//   * ring collectives straight from RCCL (all_reduce / reduce / broadcast / reduce_scatter / all_gather,
//     grouped point-to-point for pipeline stage boundaries), sub-communicators by ncclCommSplit;
//   * all_reduce_direct: the one-shot algorithm for the fully connected xGMI mesh (7 point-to-point links
//     per MI355X): every rank pushes shard j to rank j over the direct link (grouped send/recv = all links
//     busy at once), reduces its own shard locally in fp32, and pushes the result back the same way.  Wire
//     time ~ 2 * (S / N) per link instead of 2 * (N - 1) / N * S over one ring.  Optional bf16 wire format
//     (fp32 accumulation in the reduce kernel): half the bytes on every link.
//
// Links against the RCCL that torch itself loads (torch/lib/librccl.so): one RCCL per process.  The handful
// of entry points used is declared below instead of including <rccl/rccl.h>: the only header in the image
// (/opt/rocm/include/rccl, 2.27.7) is NEWER than that library (2.26.6), and these C signatures and enum
// values have been stable since NCCL 2.10, whereas struct layouts (ncclConfig_t) have not.
//
// Streams and events are torch's own (c10 stream pool, at::cuda::CUDAEvent) and every buffer handed to a
// collective is recorded on the communication stream with the caching allocator, exactly as c10d's
// ProcessGroupNCCL does - round 1's raw hipStream_t / hipEvent_t version stalled the device queues under
// bucket-sized traffic (docs/DESIGN.md section 5, "native communicator").
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

// ---- the stable C ABI of NCCL / RCCL used here ----
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int* version);
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, void* config);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
}

namespace {

using at::Tensor;
using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;

#define RCCL_CHECK(cmd) do { ncclResult_t r_ = (cmd); TORCH_CHECK(r_ == ncclSuccess, "RCCL error: ", ncclGetErrorString(r_), " at " #cmd); } while (0)

ncclDataType_t nccl_dtype(const Tensor& t) {
    switch (t.scalar_type()) {
        case at::kFloat: return ncclFloat32;
        case at::kBFloat16: return ncclBfloat16;
        case at::kHalf: return ncclFloat16;
        case at::kDouble: return ncclFloat64;
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
    static int64_t version() {
        int v = 0;
        RCCL_CHECK(ncclGetVersion(&v));
        return v;
    }

    RcclComm(const std::string& id_bytes, int rank, int world, int device)
        : rank_(rank), world_(world), device_(device), stream_(pool_stream(device)) {
        TORCH_CHECK(id_bytes.size() == sizeof(ncclUniqueId), "unique id must be ", sizeof(ncclUniqueId), " bytes");
        TORCH_CHECK(rank >= 0 && rank < world, "bad rank");
        ncclUniqueId id;
        std::memcpy(&id, id_bytes.data(), sizeof(id));
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        RCCL_CHECK(ncclCommInitRank(&comm_, world, id, rank));
    }

    ~RcclComm() {
        if (comm_ != nullptr) {
            (void)hipStreamSynchronize(stream_.stream());
            (void)ncclCommDestroy(comm_);
        }
    }

    int rank() const { return rank_; }
    int world_size() const { return world_; }

    // Sub-communicator of the ranks that pass the same `color` (< 0: this rank joins none and gets None); ranks are
    // ordered by `key`.  Collective over THIS communicator.
    std::unique_ptr<RcclComm> split(int color, int key, int new_rank, int new_world) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        // ncclCommSplit exists since NCCL 2.18; its prototype is declared by hand above, so ask the library itself
        TORCH_CHECK(version() >= 21800, "ncclCommSplit needs RCCL >= 2.18 (library reports ", version(), ")");
        ncclComm_t sub = nullptr;
        RCCL_CHECK(ncclCommSplit(comm_, color < 0 ? -1 : color, key, &sub, nullptr));   // -1 = NCCL_SPLIT_NOCOLOR
        if (color < 0 || sub == nullptr) return nullptr;
        return std::unique_ptr<RcclComm>(new RcclComm(sub, new_rank, new_world, device_));
    }

    int64_t all_reduce(Tensor t, int op) {
        check(t);
        begin();
        RCCL_CHECK(ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), nccl_dtype(t), red_op(op), comm_, stream_.stream()));
        return finish({t});
    }

    int64_t reduce(Tensor t, int root) {
        check(t);
        TORCH_CHECK(root >= 0 && root < world_, "reduce: bad root");
        begin();
        RCCL_CHECK(ncclReduce(t.data_ptr(), t.data_ptr(), t.numel(), nccl_dtype(t), ncclSum, root, comm_, stream_.stream()));
        return finish({t});
    }

    int64_t broadcast(Tensor t, int root) {
        check(t);
        TORCH_CHECK(root >= 0 && root < world_, "broadcast: bad root");
        begin();
        RCCL_CHECK(ncclBroadcast(t.data_ptr(), t.data_ptr(), t.numel(), nccl_dtype(t), root, comm_, stream_.stream()));
        return finish({t});
    }

    int64_t reduce_scatter(Tensor out, Tensor in) {
        check(out); check(in);
        TORCH_CHECK(in.numel() == out.numel() * world_ && in.scalar_type() == out.scalar_type(), "reduce_scatter: sizes");
        begin();
        RCCL_CHECK(ncclReduceScatter(in.data_ptr(), out.data_ptr(), out.numel(), nccl_dtype(out), ncclSum, comm_, stream_.stream()));
        return finish({out, in});
    }

    int64_t all_gather(Tensor out, Tensor in) {
        check(out); check(in);
        TORCH_CHECK(out.numel() == in.numel() * world_ && in.scalar_type() == out.scalar_type(), "all_gather: sizes");
        begin();
        RCCL_CHECK(ncclAllGather(in.data_ptr(), out.data_ptr(), in.numel(), nccl_dtype(in), comm_, stream_.stream()));
        return finish({out, in});
    }

    // ops: list of (is_send, tensor, peer); issued as ONE RCCL group (pipeline stage boundaries), so a send and a
    // receive between neighbours cannot deadlock
    int64_t batch_p2p(const std::vector<std::tuple<bool, Tensor, int>>& ops) {
        std::vector<Tensor> keep;
        int64_t self_send = 0, self_recv = 0;     // elements sent to / received from this rank itself inside the group
        for (auto& o : ops) {
            check(std::get<1>(o));
            const int peer = std::get<2>(o);
            TORCH_CHECK(peer >= 0 && peer < world_, "batch_p2p: bad peer ", peer);
            if (peer == rank_) (std::get<0>(o) ? self_send : self_recv) += std::get<1>(o).numel();
            keep.push_back(std::get<1>(o));
        }
        // a transfer to oneself is legal only as a matched send + receive inside ONE group (RCCL then copies locally);
        // a lone one would wait for a peer that cannot arrive
        TORCH_CHECK(self_send == self_recv, "batch_p2p: a send to this rank itself needs the matching receive in the same group");
        begin();
        RCCL_CHECK(ncclGroupStart());
        for (auto& o : ops) {
            const Tensor& t = std::get<1>(o);
            const int peer = std::get<2>(o);
            if (std::get<0>(o)) RCCL_CHECK(ncclSend(t.data_ptr(), t.numel(), nccl_dtype(t), peer, comm_, stream_.stream()));
            else                RCCL_CHECK(ncclRecv(t.data_ptr(), t.numel(), nccl_dtype(t), peer, comm_, stream_.stream()));
        }
        RCCL_CHECK(ncclGroupEnd());
        return finish(keep);
    }

    // One-shot all-reduce (sum) of an fp32 buffer for a fully connected topology.  `scratch` (fp32, or bf16 for the
    // bf16 wire format) holds the (world - 1) incoming copies of this rank's shard, and with bf16 also this rank's
    // outgoing casts: numel >= (world - 1) * ceil(n / world) [fp32] or (world + world - 1) * ceil(n / world) [bf16].
    int64_t all_reduce_direct(Tensor t, Tensor scratch) {
        check(t); check(scratch);
        TORCH_CHECK(t.scalar_type() == at::kFloat, "all_reduce_direct: fp32 buffer");
        const bool bf16 = scratch.scalar_type() == at::kBFloat16;
        TORCH_CHECK(bf16 || scratch.scalar_type() == at::kFloat, "all_reduce_direct: scratch must be fp32 or bf16");
        const int64_t n = t.numel();
        const int64_t s = ((n + world_ - 1) / world_ + 7) / 8 * 8;          // shard length, 16-byte aligned in bf16
        const int64_t need = bf16 ? (int64_t)(2 * world_ - 1) * s : (int64_t)(world_ - 1) * s;
        TORCH_CHECK(scratch.numel() >= need, "all_reduce_direct: scratch too small (", scratch.numel(), " < ", need, ")");
        auto lo = [&](int r) { return std::min<int64_t>(n, (int64_t)r * s); };
        auto len = [&](int r) { return std::min<int64_t>(n, (int64_t)(r + 1) * s) - lo(r); };
        float* base = t.data_ptr<float>();
        hipStream_t st = stream_.stream();
        begin();
        if (world_ > 1) {
            const int64_t mine = len(rank_);
            if (!bf16) {
                float* scr = scratch.data_ptr<float>();
                // phase 1: shard j -> rank j over the direct link; everyone's copy of MY shard arrives in scratch
                RCCL_CHECK(ncclGroupStart());
                int slot = 0;
                for (int p = 0; p < world_; ++p) {
                    if (p == rank_) continue;
                    if (len(p) > 0) RCCL_CHECK(ncclSend(base + lo(p), len(p), ncclFloat32, p, comm_, st));
                    if (mine > 0) RCCL_CHECK(ncclRecv(scr + (int64_t)slot * s, mine, ncclFloat32, p, comm_, st));
                    ++slot;
                }
                RCCL_CHECK(ncclGroupEnd());
                // local N-way reduction in fp32 on the comm stream
                if (mine > 0) dpa::reduce_shards(base + lo(rank_), scr, world_ - 1, s, mine, st);
                // phase 2: the reduced shard goes back to everyone, theirs arrive in place
                RCCL_CHECK(ncclGroupStart());
                for (int p = 0; p < world_; ++p) {
                    if (p == rank_) continue;
                    if (mine > 0) RCCL_CHECK(ncclSend(base + lo(rank_), mine, ncclFloat32, p, comm_, st));
                    if (len(p) > 0) RCCL_CHECK(ncclRecv(base + lo(p), len(p), ncclFloat32, p, comm_, st));
                }
                RCCL_CHECK(ncclGroupEnd());
            } else {
                // bf16 on the wire, fp32 accumulation: [0, world * s) of scratch = bf16 image of the whole buffer
                // (outgoing shards; later the reduced shards coming back), the rest = incoming copies of my shard
                uint16_t* out = reinterpret_cast<uint16_t*>(scratch.data_ptr());
                uint16_t* in = out + (int64_t)world_ * s;
                dpa::cast_f32_to_bf16(base, out, n, st);
                RCCL_CHECK(ncclGroupStart());
                int slot = 0;
                for (int p = 0; p < world_; ++p) {
                    if (p == rank_) continue;
                    if (len(p) > 0) RCCL_CHECK(ncclSend(out + lo(p), len(p), ncclBfloat16, p, comm_, st));
                    if (mine > 0) RCCL_CHECK(ncclRecv(in + (int64_t)slot * s, mine, ncclBfloat16, p, comm_, st));
                    ++slot;
                }
                RCCL_CHECK(ncclGroupEnd());
                // my shard: own fp32 values + the peers' bf16 copies, summed in fp32; result in fp32 (in place) and bf16 (to send)
                if (mine > 0) dpa::reduce_shards_bf16(base + lo(rank_), in, world_ - 1, s, mine, out + lo(rank_), st);
                RCCL_CHECK(ncclGroupStart());
                for (int p = 0; p < world_; ++p) {
                    if (p == rank_) continue;
                    if (mine > 0) RCCL_CHECK(ncclSend(out + lo(rank_), mine, ncclBfloat16, p, comm_, st));
                    if (len(p) > 0) RCCL_CHECK(ncclRecv(out + lo(p), len(p), ncclBfloat16, p, comm_, st));
                }
                RCCL_CHECK(ncclGroupEnd());
                // the peers' reduced shards back to fp32, in place; mine is already there
                for (int p = 0; p < world_; ++p)
                    if (p != rank_ && len(p) > 0) dpa::cast_bf16_to_f32(out + lo(p), base + lo(p), len(p), st);
            }
        }
        return finish({t, scratch});
    }

    // the CURRENT stream waits for the operation behind `ticket` (no host block)
    // A ticket may be waited for from SEVERAL streams (the optimizer's stream behind one bucket, the compute stream in
    // DataParallel.finish()): the event stays until it has completed, finish() prunes completed ones.
    void wait(int64_t ticket) {
        auto it = pending_.find(ticket);
        if (it == pending_.end()) return;                     // completed and pruned
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        Stream cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_);
        it->second->block(cur);
    }

    // host-side: is the operation behind `ticket` complete? (watchdog / tests)
    bool query(int64_t ticket) {
        auto it = pending_.find(ticket);
        return it == pending_.end() ? true : it->second->query();
    }

    void synchronize() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        C10_HIP_CHECK(hipStreamSynchronize(stream_.stream()));
        pending_.clear();
    }

    void abort() {
        if (comm_ != nullptr) { (void)ncclCommAbort(comm_); comm_ = nullptr; }
    }

  private:
    RcclComm(ncclComm_t comm, int rank, int world, int device)
        : rank_(rank), world_(world), device_(device), comm_(comm), stream_(pool_stream(device)) {}

    static Stream pool_stream(int device) {
        return c10::hip::getStreamFromPoolMasqueradingAsCUDA(/*isHighPriority=*/true, (c10::DeviceIndex)device);
    }
    c10::Device dev() const { return c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device_); }

    void check(const Tensor& t) const {
        TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "RcclComm: contiguous device tensors only");
        TORCH_CHECK(t.device().index() == device_, "RcclComm: tensor on device ", (int)t.device().index(), ", communicator on ", device_);
    }
    static ncclRedOp_t red_op(int op) { return op == 1 ? ncclMax : op == 2 ? ncclMin : ncclSum; }

    // comm stream waits for everything enqueued on the CURRENT stream so far (the producer of the buffer)
    void begin() {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev());
        Stream cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_);
        ready_.record(cur);
        ready_.block(stream_);
    }

    // completion event on the comm stream; the buffers are recorded on that stream with the caching allocator, so
    // their memory is not handed out again before the collective has passed (the caller need not keep them alive)
    int64_t finish(const std::vector<Tensor>& used) {
        for (const Tensor& t : used)
            c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(t.storage().data_ptr(), stream_);
        for (auto it = pending_.begin(); it != pending_.end();)          // operations that have completed need no event any more
            it = it->second->query() ? pending_.erase(it) : std::next(it);
        auto done = std::make_unique<at::cuda::CUDAEvent>(hipEventDisableTiming);
        done->record(stream_);
        const int64_t ticket = next_ticket_++;
        pending_[ticket] = std::move(done);
        return ticket;
    }

    int rank_, world_, device_;
    ncclComm_t comm_ = nullptr;
    Stream stream_;
    at::cuda::CUDAEvent ready_{hipEventDisableTiming};
    std::map<int64_t, std::unique_ptr<at::cuda::CUDAEvent>> pending_;
    int64_t next_ticket_ = 1;
};

}  // namespace

void bind_rccl(pybind11::module_& m) {
    py::class_<RcclComm>(m, "RcclComm")
        .def(py::init<const std::string&, int, int, int>(), py::arg("unique_id"), py::arg("rank"), py::arg("world_size"), py::arg("device"))
        .def_static("unique_id", &RcclComm::unique_id)
        .def_static("version", &RcclComm::version)
        .def_property_readonly("rank", &RcclComm::rank)
        .def_property_readonly("world_size", &RcclComm::world_size)
        .def("split", &RcclComm::split, py::arg("color"), py::arg("key"), py::arg("new_rank"), py::arg("new_world"))
        .def("all_reduce", &RcclComm::all_reduce, py::arg("tensor"), py::arg("op") = 0)
        .def("all_reduce_direct", &RcclComm::all_reduce_direct)
        .def("reduce", &RcclComm::reduce)
        .def("broadcast", &RcclComm::broadcast)
        .def("reduce_scatter", &RcclComm::reduce_scatter)
        .def("all_gather", &RcclComm::all_gather)
        .def("batch_p2p", &RcclComm::batch_p2p)
        .def("wait", &RcclComm::wait)
        .def("query", &RcclComm::query)
        .def("synchronize", &RcclComm::synchronize)
        .def("abort", &RcclComm::abort);
}
