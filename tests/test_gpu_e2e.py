"""T3: single-GPU end-to-end through the full engine (flat buffers -> bucket -> self
all-reduce -> fused SGD), fused-vs-unfused block parity, checkpoint round trip on GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def two_pass_shortcut():
    """Tests that hold the fused blocks to the op-by-op path at a tight bound pin the two-pass projection shortcut:
    the folded apply (ops/block.py fold_ds, default) skips one bf16 rounding of the op-by-op forward, so ReLU bits next
    to zero differ and gradients agree only in aggregate - it is held to its own bound in
    test_folded_shortcut_apply_matches_two_pass_block."""
    from distributedpytorch_amd.ops import block
    keep = block._FOLD_DS_APPLY
    block._FOLD_DS_APPLY = False
    yield
    block._FOLD_DS_APPLY = keep


def _train(model, x, y, steps, loss_fn=None, use_lp=True, lr=0.05):
    from distributedpytorch_amd.parallel.dp import DataParallel
    from distributedpytorch_amd.optim import FusedSGD
    dp = DataParallel(model, use_lp=use_lp)
    opt = FusedSGD(dp.flat, lr=lr, momentum=0.9)
    losses = []
    for _ in range(steps):
        loss = loss_fn(model, x, y) if loss_fn else model.loss(x, y)
        loss.backward()
        dp.finish()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, dp, opt


def test_small_cnn_trains_through_engine_and_checkpoint(tmp_path):
    from distributedpytorch_amd.models import mnist
    from distributedpytorch_amd import checkpoint as ck
    torch.manual_seed(0)
    model = mnist.Net().to(DEV).train()
    from distributedpytorch_amd.data import SyntheticImageDataset
    x, y = SyntheticImageDataset(256, (1, 28, 28), 10, seed=3).batch(range(128))     # learnable: class prototypes + noise
    x, y = x.to(DEV), y.to(DEV)
    losses, dp, opt = _train(model, x, y, 40, lambda m, a, b: mnist.nll_loss(m(a), b), use_lp=False)
    assert losses[-1] < 0.7 * losses[0], losses
    path = str(tmp_path / "net.pt")
    ck.save_checkpoint(path, model, opt, 1)
    model.eval()
    ref = model(x).clone()
    m2 = mnist.Net().to(DEV).eval()
    ck.load_checkpoint(path, m2)
    assert torch.allclose(m2(x), ref, atol=1e-5)
    # a stock-PyTorch user loads the file with plain torch.load
    sd = torch.load(path + ".model", weights_only=True)
    assert sd["conv1.weight"].shape == (10, 1, 5, 5) and sd["fc2.bias"].dtype == torch.float32


def test_fused_block_matches_op_by_op_path(two_pass_shortcut):
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import block
    from distributedpytorch_amd.parallel.flat import FlatParams
    x = torch.randn(8, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (8,), device=DEV)
    grads = []
    for enabled in (True, False, "recompute"):
        torch.manual_seed(1)
        m = resnet.tiny_resnet().to(DEV).train()
        for mod in m.modules():      # non-trivial BN affine (bn3 gamma is zero-initialised)
            if hasattr(mod, "running_mean"):
                torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0.0, 0.2)
        flat = FlatParams(m)
        flat.install_sink(None)
        block.ENABLED = bool(enabled)
        keep = block.RECOMPUTE_MIN_BYTES
        block.RECOMPUTE_MIN_BYTES = 0 if enabled == "recompute" else 1 << 60   # force / forbid conv3+bn3 recomputation
        try:
            loss = m.loss(x, y)
            loss.backward()
        finally:
            block.ENABLED = True
            block.RECOMPUTE_MIN_BYTES = keep
        grads.append((float(loss.detach()), flat.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    for (l1, g1, d1), (l2, g2, d2) in ((grads[0], grads[1]), (grads[2], grads[1])):
        _compare_paths(l1, g1, d1, l2, g2, d2)


def test_folded_shortcut_apply_matches_two_pass_block():
    """Projection-shortcut BatchNorm applied inside bn3's apply kernel (ops/block.py fold_ds) against the two-pass form
    on one downsample block: outputs agree within the bf16 rounding of the shortcut the fold removes, ReLU bits differ
    only where the pre-activation is within that rounding of zero, and with the SAME output gradient the parameter
    and input gradients agree up to those bits."""
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import block
    from distributedpytorch_amd.parallel.flat import FlatParams
    torch.manual_seed(7)
    x0 = (torch.randn(8, 28, 28, 64, device=DEV) * 0.7).to(torch.bfloat16)
    dout = (torch.randn(8, 14, 14, 128, device=DEV) * 0.1).to(torch.bfloat16)
    got = []
    keep = block._FOLD_DS_APPLY
    try:
        for fold in (True, False):
            torch.manual_seed(8)
            blk = resnet.Bottleneck(64, 32, stride=2, downsample=True).to(DEV).train()
            for mod in blk.modules():
                if hasattr(mod, "running_mean"):
                    torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0.0, 0.2)
            flat = FlatParams(blk)
            block._FOLD_DS_APPLY = fold
            x = x0.clone().requires_grad_(True)
            assert block.can_fuse(blk, x)
            out = blk(x)
            out.backward(dout)
            torch.cuda.synchronize()
            got.append((out.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in blk.named_parameters()},
                        blk.downsample[1].running_mean.clone()))
    finally:
        block._FOLD_DS_APPLY = keep
    (o1, dx1, g1, rm1), (o2, dx2, g2, rm2) = got
    assert torch.equal(rm1, rm2)                                   # same statistics, same running-stat update
    _ = (o1 - o2).abs()
    assert float(_.max()) <= 0.05 and float(_.mean()) < 2e-3, (float(_.max()), float(_.mean()))
    flipped = (o1 > 0) != (o2 > 0)
    assert float(flipped.float().mean()) < 5e-3
    near = torch.maximum(o1, o2)[flipped]
    assert near.numel() == 0 or float(near.max()) < 0.05              # flips only next to zero
    cos = torch.nn.functional.cosine_similarity(dx1.flatten(), dx2.flatten(), dim=0).item()
    assert cos > 0.995, cos
    for n in g1:
        assert (g1[n] - g2[n]).norm().item() <= 0.1 * g2[n].norm().item() + 1e-5, n


def test_side_stream_wgrad_matches_in_line_wgrad():
    """Weight gradients launched on the side stream (ops/side.py) equal the in-line ones, and are complete when
    backward() returns (end-of-backward join) as well as after DataParallel.finish()."""
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import side
    from distributedpytorch_amd.parallel.dp import DataParallel
    x = torch.randn(16, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (16,), device=DEV)
    got = []
    keep = side.ENABLED
    try:
        for enabled in (True, False):
            side.ENABLED = enabled
            torch.manual_seed(4)
            m = resnet.tiny_resnet().to(DEV).train()
            dp = DataParallel(m)
            m.loss(x, y).backward()
            g_after_backward = dp.flat.grad.clone()
            dp.finish()
            got.append((g_after_backward, dp.flat.grad.clone()))
    finally:
        side.ENABLED = keep
    (a0, a1), (b0, b1) = got
    assert torch.equal(a0, a1) and torch.equal(b0, b1)          # nothing was still in flight after backward()
    cos = torch.nn.functional.cosine_similarity(a1, b1, dim=0).item()
    assert cos > 0.999 and ((a1 - b1).norm() / b1.norm()).item() < 0.05, cos


def test_two_engines_in_one_process_each_fill_their_own_flat_buffer():
    """Two DataParallel wrappers alive at once: the gradient sink hangs off the parameters, so the
    second engine does not steal the first one's kernels' output - interleaved backward passes land in the right
    flat buffers and equal what each model produces alone."""
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.parallel.dp import DataParallel
    x = torch.randn(8, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (8,), device=DEV)

    def build(seed):
        torch.manual_seed(seed)
        m = resnet.tiny_resnet().to(DEV).train()
        return m, DataParallel(m)

    solo = []
    for seed in (11, 12):
        m, dp = build(seed)
        m.loss(x, y).backward(); dp.finish()
        solo.append(dp.flat.grad.clone())
    (m1, dp1), (m2, dp2) = build(11), build(12)          # both alive, both with a sink installed
    l1, l2 = m1.loss(x, y), m2.loss(x, y)
    l2.backward(); l1.backward()
    dp1.finish(); dp2.finish()
    for got, want in ((dp1.flat.grad, solo[0]), (dp2.flat.grad, solo[1])):
        assert float(got.norm()) > 0
        assert ((got - want).norm() / want.norm()).item() < 2e-2       # atomics' summation order only


def test_block_output_with_a_second_consumer_falls_back_to_standalone_bn3_statistics(two_pass_shortcut):
    """The next block folds bn3's backward reduction into its last dgrad only if bn3's incoming gradient IS the
    tensor that dgrad wrote.  With an auxiliary consumer of the block output autograd sums two gradients: the folded
    statistics must be discarded (and the accumulator re-zeroed) - gradients equal the op-by-op path."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import block
    from distributedpytorch_amd.parallel.flat import FlatParams
    x = torch.randn(8, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (8,), device=DEV)
    res = []
    for enabled in (True, False):
        torch.manual_seed(6)
        m = resnet.tiny_resnet().to(DEV).train()
        for mod in m.modules():
            if hasattr(mod, "running_mean"):
                torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0.0, 0.2)
        flat = FlatParams(m)
        flat.install_sink(None)
        block.ENABLED = enabled
        try:
            for _ in range(2):                     # twice: a dirty accumulator would corrupt the second step
                flat.grad.zero_()
                y1 = m.block1(m.stem(x))
                logits = m.head(m.block2(y1))
                loss = ops.softmax_cross_entropy(logits, y, 16) + 0.05 * y1.float().square().mean()
                loss.backward()
        finally:
            block.ENABLED = True
        res.append((float(loss.detach()), flat.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    _compare_paths(*res[0], *res[1])


def _compare_paths(l1, g1, d1, l2, g2, d2):
    assert abs(l1 - l2) < 1e-3
    # The BN statistics are summed with fp32 atomics (order varies run to run), so an activation
    # within one bf16 rounding of zero can land on either side of a ReLU mask: single elements of
    # small gradients may differ visibly between two runs of the SAME path.  Compare in aggregate.
    cos = torch.nn.functional.cosine_similarity(g1, g2, dim=0).item()
    rel = ((g1 - g2).norm() / g2.norm()).item()
    assert cos > 0.999 and rel < 0.05, (cos, rel)
    for n in d1:
        a, b = d1[n], d2[n]
        assert (a - b).norm().item() <= 0.1 * b.norm().item() + 1e-6, n


def test_tiny_resnet_trains_and_matches_cpu_first_step():
    from distributedpytorch_amd.models import resnet
    torch.manual_seed(2)
    m = resnet.tiny_resnet().to(DEV).train()
    x = torch.randn(32, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (32,), device=DEV)
    torch.manual_seed(2)
    ref = resnet.tiny_resnet().train()
    lref = float(ref.loss(x.cpu(), y.cpu()))
    losses, dp, opt = _train(m, x, y, 12)
    assert abs(losses[0] - lref) < 0.05, (losses[0], lref)          # bf16 forward vs fp32 CPU oracle
    assert losses[-1] < losses[0]
    assert dp.stats["bucket_launches"] >= 12
    assert dp.stats["allreduce_calls"] == 0 and dp.stats["allreduce_bytes"] == 0    # one process, no group: nothing was reduced


def test_gpt2_tiny_trains_on_gpu():
    from distributedpytorch_amd.ops import _ext
    if not hasattr(_ext.load(), "attention_fwd"):
        pytest.skip("attention kernels not built")
    from distributedpytorch_amd.models import gpt2
    torch.manual_seed(3)
    m = gpt2.gpt2_tiny(n_layer=2, n_embd=128, n_head=2, vocab=500, n_positions=128).to(DEV).train()
    tok = torch.randint(0, 500, (4, 128), device=DEV); tgt = torch.randint(0, 500, (4, 128), device=DEV)
    torch.manual_seed(3)
    ref = gpt2.gpt2_tiny(n_layer=2, n_embd=128, n_head=2, vocab=500, n_positions=128).train()
    lref = float(ref.loss(tok.cpu(), tgt.cpu()))
    losses, _, _ = _train(m, tok, tgt, 10, lr=0.02)
    assert abs(losses[0] - lref) < 0.05, (losses[0], lref)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("which", ["tiny_resnet", "gpt2_tiny"])
def test_optimizer_overlapped_with_backward_follows_the_same_trajectory(which):
    """``FusedSGD.overlap_with_backward``: buckets updated on their own stream behind their reduction, while backward is
    still running, must give the parameters the single end-of-step launch gives (same arithmetic, earlier).  Small
    buckets so that several updates really happen inside backward; momentum + weight decay + per-segment decay table."""
    from distributedpytorch_amd.optim import FusedSGD
    from distributedpytorch_amd.parallel.dp import DataParallel

    def run(overlap):
        torch.manual_seed(7)
        if which == "tiny_resnet":
            from distributedpytorch_amd.models import resnet
            m = resnet.tiny_resnet().to(DEV).train()
            g = torch.Generator(device=DEV).manual_seed(3)
            x = torch.randn(16, 3, 64, 64, generator=g, device=DEV)
            y = torch.randint(0, 16, (16,), generator=g, device=DEV)
        else:
            from distributedpytorch_amd.models import gpt2
            m = gpt2.gpt2_tiny(n_layer=2, n_embd=128, n_head=2, vocab=504, n_positions=128).to(DEV).train()
            g = torch.Generator(device=DEV).manual_seed(3)
            x = torch.randint(0, 504, (2, 128), generator=g, device=DEV)
            y = torch.randint(0, 504, (2, 128), generator=g, device=DEV)
        dp = DataParallel(m, bucket_mb=0.05, first_bucket_mb=0.02)
        opt = FusedSGD(dp.flat, lr=0.05, momentum=0.9, weight_decay=1e-3)
        opt.grad_scale = dp.grad_scale
        assert opt.overlap_with_backward(dp, enabled=overlap) == overlap
        assert len(dp.buckets) >= 3
        losses = []
        for _ in range(5):
            loss = m.loss(x, y)
            loss.backward()
            dp.finish()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        assert float(dp.flat.grad.abs().max()) == 0.0                     # every gradient consumed and zeroed
        return losses, dp.flat.master.clone(), opt.momentum_buf.clone()

    # The comparison runs in DETERMINISTIC mode (no float atomics: csrc/runtime.hip), where the overlapped and the plain
    # step must agree bit for bit - same arithmetic, only earlier.  With the default atomic kernels two PLAIN runs of this
    # tiny model land on one of a few discrete trajectories (a sum that differs in its last bit flips one bf16 rounding;
    # BatchNorm amplifies it: parameter distance 1e-7 three times out of four, 2e-4 .. 1e-3 otherwise -
    # scripts/flaky_overlap.py), so "as close as two plain runs" was a coin that came up tails in one run out of six.
    import distributedpytorch_amd as dpa
    dpa.set_deterministic(True)
    try:
        l0, p0, m0 = run(False)
        l1, p1, m1 = run(True)
    finally:
        dpa.set_deterministic(False)
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())
    assert torch.equal(m0, m1), float((m0 - m1).abs().max())
    assert l0 == l1
    # ... and with the default (atomic) kernels the overlapped step stays on one of those trajectories
    l2, p2, m2 = run(True)
    assert float((p0 - p2).abs().max()) <= 5e-3, float((p0 - p2).abs().max())
    assert max(abs(a - b) for a, b in zip(l0, l2)) <= 5e-3


def test_mnist_hipgraph_step_matches_eager_and_draws_fresh_masks():
    """Forward+backward captured in a hipGraph: gradients equal the eager step's (same device-counter
    dropout stream), and a second replay draws different dropout masks."""
    from distributedpytorch_amd.models import mnist
    from distributedpytorch_amd.parallel.dp import DataParallel
    from distributedpytorch_amd.parallel.graph import GraphedStep
    from distributedpytorch_amd.data import synthetic_batch
    x, y = synthetic_batch("mnist", 64, DEV, seed=5)

    def build():
        torch.manual_seed(1234)
        m = mnist.Net().to(DEV).train()
        m._drop.use_device_counter(DEV)
        return m, DataParallel(m, use_lp=False)

    m1, dp1 = build()
    grads_eager = []
    for _ in range(2):
        dp1.flat.grad.zero_()
        with dp1.no_sync():
            m1.loss(x, y).backward()                # the head kernel bumps the device dropout counter
        grads_eager.append(dp1.flat.grad.clone())
    m2, dp2 = build()
    g = GraphedStep(dp2, lambda a, b: m2.loss(a, b), (x, y), warmup=2)
    m2._drop.dev_step.zero_()                       # the warm-up advanced the counter; restart the stream
    grads_graph = []
    for _ in range(2):
        dp2.flat.grad.zero_()
        loss = g(x, y)
        grads_graph.append(dp2.flat.grad.clone())
    assert torch.isfinite(loss).all()
    for a, b in zip(grads_eager, grads_graph):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    assert not torch.equal(grads_graph[0], grads_graph[1])     # fresh dropout masks per replay
    assert int(m2._drop.dev_step.item()) == 2


def test_native_rccl_communicator_single_rank():
    """csrc/rccl_comm.cpp at world size 1 (all a one-GPU box allows): bootstrap, comm-stream ordering against
    the compute stream, ring and direct all-reduce, broadcast, reduce-scatter / all-gather, ticket waits."""
    from distributedpytorch_amd.ops import _ext
    C = _ext.load()
    nc = C.RcclComm(C.RcclComm.unique_id(), 0, 1, torch.cuda.current_device())
    assert nc.rank == 0 and nc.world_size == 1
    x = torch.randn(1 << 20, device=DEV)
    ref = x.clone()
    y = x * 2.0                                   # producer on the compute stream; the comm stream must wait for it
    t = nc.all_reduce(y, 0)
    nc.wait(t)
    z = y + 1.0                                   # consumer on the compute stream, ordered behind the ticket
    torch.testing.assert_close(z, ref * 2.0 + 1.0)
    scratch = torch.empty(0, device=DEV)
    nc.wait(nc.all_reduce_direct(y, scratch))
    torch.testing.assert_close(y, ref * 2.0)
    nc.wait(nc.broadcast(y, 0))
    out = torch.empty_like(y)
    nc.wait(nc.reduce_scatter(out, y)); torch.testing.assert_close(out, y)
    out.zero_(); nc.wait(nc.all_gather(out, y)); torch.testing.assert_close(out, y)
    nc.wait(nc.reduce(y, 0)); torch.testing.assert_close(y, ref * 2.0)
    nc.wait(nc.all_reduce_direct(y, torch.empty(1 << 20, dtype=torch.bfloat16, device=DEV)))      # bf16 wire format
    torch.testing.assert_close(y, ref * 2.0)
    t = nc.all_reduce(y, 0)
    nc.synchronize()
    assert nc.query(t) and C.RcclComm.version() >= 21000
    # the N-way reduction kernel of the direct algorithm: dst += sum of (world - 1) incoming shards
    dst = torch.randn(1003, device=DEV); src = torch.randn(3, 1024, device=DEV)
    want = dst + src[:, :1003].sum(0)
    C.reduce_shards(dst, src.view(-1), 3, 1024)
    torch.testing.assert_close(dst, want)
    # ... and its bf16-wire twin: fp32 accumulation of bf16 copies, result in fp32 and in bf16
    dst = torch.randn(1000, device=DEV); srcb = torch.randn(3, 1024, device=DEV).to(torch.bfloat16)
    want = dst + srcb[:, :1000].float().sum(0)
    outb = torch.empty(1000, dtype=torch.bfloat16, device=DEV)
    C.reduce_shards_bf16(dst, srcb.view(-1), 3, 1024, outb)
    assert torch.equal(outb, want.to(torch.bfloat16))
    # the owner keeps exactly what its peers will widen from the wire: replicas stay bit-identical
    assert torch.equal(dst, outb.float())
    torch.testing.assert_close(dst, want, rtol=8e-3, atol=1e-3)


def _self_exchange_worker(local_rank, global_rank, world_size, out_dir):
    import os
    import time
    import torch
    from distributedpytorch_amd.ops import _ext
    C = _ext.load()
    torch.cuda.set_device(0)
    nc = C.RcclComm(C.RcclComm.unique_id(), 0, 1, 0)
    a = (torch.arange(1 << 20, device="cuda", dtype=torch.float32) * 0.5).to(torch.bfloat16)
    src = a * 2                                          # producer on the compute stream
    dst = torch.zeros_like(src)
    ticket = nc.batch_p2p([(True, src, 0), (False, dst, 0)])
    t0 = time.monotonic()
    while not nc.query(ticket):
        if time.monotonic() - t0 > 30.0:
            nc.abort()
            raise RuntimeError("grouped self send/recv did not complete in 30 s")
        time.sleep(0.001)
    nc.wait(ticket)
    ok = bool(torch.equal(dst + 0, a * 2))               # consumer on the compute stream, ordered behind the ticket
    refused = False
    try:
        nc.batch_p2p([(True, src, 0)])                   # a lone transfer to oneself could never complete: refused
    except RuntimeError as e:
        refused = "matching receive" in str(e)
    nc.synchronize()
    torch.save({"ok": ok, "refused": refused}, os.path.join(out_dir, "self.pt"))


def test_native_grouped_send_recv_runs_on_one_gpu_as_a_self_exchange(tmp_path):
    """ncclSend / ncclRecv through ``RcclComm.batch_p2p`` on the one GPU a box has: a rank may exchange with ITSELF
    when the send and the matching receive sit in one group (RCCL then copies locally) - the same C++ path, group
    calls, comm-stream ordering and ticket hand-off that carry the pipeline's stage boundaries between GPUs.  Under a
    host watchdog, and in a process of its own: RCCL's self-transfer path is not one the training runs ever take, and a
    communicator created in the same process right after it stalled under bucket-sized traffic on one box."""
    from distributedpytorch_amd import launch
    launch.spawn(_self_exchange_worker, [1], 0, args=(str(tmp_path),))
    r = torch.load(tmp_path / "self.pt", weights_only=True)
    assert r["ok"] and r["refused"]


def test_native_rccl_layer_under_bucket_sized_traffic():
    """The pattern that stalled round 1's raw-stream version of the communicator (docs/DESIGN.md section 5): five
    32 MB in-place all-reduces per step, each issued from a side stream behind a producer kernel, all waited for on
    the compute stream, then a host read - 30 steps, with a host watchdog instead of a hang."""
    import time
    from distributedpytorch_amd.ops import _ext
    C = _ext.load()
    nc = C.RcclComm(C.RcclComm.unique_id(), 0, 1, torch.cuda.current_device())
    bufs = [torch.zeros(8 << 20, device=DEV) for _ in range(5)]
    side = torch.cuda.Stream()
    for step in range(30):
        tickets = []
        for b in bufs:
            b.add_(1.0)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                b.mul_(1.0)
                tickets.append(nc.all_reduce(b, 0))
        for t in tickets:
            nc.wait(t)
        t0 = time.monotonic()
        while not all(nc.query(t) for t in tickets):
            assert time.monotonic() - t0 < 30.0, f"native communicator stalled in step {step}"
            time.sleep(0.001)
        assert sum(float(b[0]) for b in bufs) == 5.0 * (step + 1)
    nc.synchronize()


def test_native_group_single_rank_carries_the_dp_engine(monkeypatch):
    """init_native_group (store rendezvous, no torch.distributed) at world size 1 with the collectives forced through
    RCCL: the DataParallel engine's bucket launches, side-stream issue and optimizer ordering on the native layer."""
    from distributedpytorch_amd import comm as dcomm
    from distributedpytorch_amd.models.resnet import tiny_resnet
    from distributedpytorch_amd.parallel.dp import DataParallel
    from distributedpytorch_amd.optim import FusedSGD
    from distributedpytorch_amd.utils import testing as T
    monkeypatch.setattr(dcomm, "_FORCE_COLLECTIVES", True)
    c = dcomm.init_native_group(f"tcp://127.0.0.1:{T.free_port()}", 0, 1, torch.device(DEV, torch.cuda.current_device()))
    try:
        assert c.native is not None and c.world_size == 1 and c.backend == "nccl"
        torch.manual_seed(0)
        model = tiny_resnet().to(DEV).train()
        dp = DataParallel(model, c, bucket_mb=0.25, first_bucket_mb=0.0625)
        opt = FusedSGD(dp.flat, lr=0.05, momentum=0.9)
        x = torch.randn(16, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (16,), device=DEV)
        losses = []
        for _ in range(6):
            loss = model.loss(x, y); loss.backward(); dp.finish(); opt.step()
            losses.append(float(loss))
        assert dp.stats["allreduce_calls"] >= 6 * len(dp.buckets) and len(dp.buckets) > 1
        assert losses[-1] < losses[0]
        assert c.all_reduce_scalar(3.0, "max") == 3.0 and c.broadcast_object({"a": 1}) == {"a": 1}
        c.barrier()
    finally:
        c.close()


@pytest.mark.parametrize("model_name", ["mnist", "tiny_resnet"])
def test_two_ranks_share_the_gpu_and_match_the_cpu_reference(tmp_path, model_name):
    """World size 2 on a one-GPU box: both ranks run the HIP engine on GPU 0 (wgrad side stream, hook-driven bucket
    launches, fused optimizer) and all-reduce over gloo.  Mean-of-gradients semantics against a CPU fp32 oracle, and
    bit-identical replicas after 3 optimizer steps."""
    from distributedpytorch_amd import launch
    from distributedpytorch_amd.utils import testing as T
    world = 2
    url = f"tcp://127.0.0.1:{T.free_port()}"
    launch.spawn(T.dp_worker, [world], 0, args=(url, model_name, str(tmp_path), 64, 3, "hooks", "gloo_gpu"))
    model, loss_fn, shape, classes = T._make_model(model_name)
    x, y = T._data(shape, classes, 8 * world)
    acc = None
    for r in range(world):
        model.zero_grad()
        loss_fn(model, x[r * 8:(r + 1) * 8], y[r * 8:(r + 1) * 8]).backward()
        g = {n: p.grad.clone() for n, p in model.named_parameters()}
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    ref = {n: v / world for n, v in acc.items()}
    got = [torch.load(tmp_path / f"grads_{r}.pt", weights_only=True) for r in range(world)]
    for name in got[0]:
        assert torch.equal(got[0][name], got[1][name]), f"ranks disagree on the reduced gradient of {name}"
    if model_name == "mnist":                       # fp32 kernels: tight; the bf16 ResNet is compared in aggregate
        for name, g in ref.items():
            assert torch.allclose(got[0][name], g, rtol=2e-3, atol=2e-5), name
    finals = [torch.load(tmp_path / f"final_{r}.pt", weights_only=True) for r in range(world)]
    assert finals[0]["calls"] >= 3
    for name in finals[0]["params"]:
        assert torch.equal(finals[0]["params"][name], finals[1]["params"][name]), f"replicas diverged at {name}"


def _rccl_world1_worker(local_rank, global_rank, world_size, url, out_dir):
    import os
    import torch
    from distributedpytorch_amd import comm as dcomm
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.optim import FusedSGD
    from distributedpytorch_amd.parallel.dp import DataParallel
    dcomm._FORCE_COLLECTIVES = True                    # what DPA_FORCE_COLLECTIVES=1 sets at import
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm = dcomm.init_process_group("nccl", url, 0, 1, 120.0, dev)
    torch.manual_seed(11)
    m = resnet.tiny_resnet().to(dev).train()
    x = torch.randn(16, 3, 64, 64, device=dev); y = torch.randint(0, 16, (16,), device=dev)
    dp = DataParallel(m, comm, bucket_mb=0.05, first_bucket_mb=0.01)
    opt = FusedSGD(dp.flat, lr=0.05, momentum=0.9)
    losses = []
    for _ in range(6):
        loss = m.loss(x, y); loss.backward(); dp.finish(); opt.step()
        losses.append(float(loss.detach()))
    comm.barrier()
    torch.save({"losses": losses, "calls": dp.stats["allreduce_calls"], "buckets": len(dp.buckets)}, os.path.join(out_dir, "r.pt"))
    dcomm.destroy_process_group()


def test_dp_engine_over_real_rccl_collectives_at_world_size_one(tmp_path):
    """The N-GPU call path on one GPU: c10d's RCCL process group (one rank), every bucket all-reduced for real on
    RCCL's stream during backward, side-stream joins in front of each launch, optimizer behind the waits.  The
    losses must equal a run without any communicator."""
    from distributedpytorch_amd import launch
    from distributedpytorch_amd.utils import testing as T
    url = f"tcp://127.0.0.1:{T.free_port()}"
    launch.spawn(_rccl_world1_worker, [1], 0, args=(url, str(tmp_path)))
    r = torch.load(tmp_path / "r.pt", weights_only=True)
    assert r["buckets"] >= 2 and r["calls"] >= 6 * r["buckets"]
    torch.manual_seed(11)
    from distributedpytorch_amd.models import resnet
    m = resnet.tiny_resnet().to(DEV).train()
    x = torch.randn(16, 3, 64, 64, device=DEV); y = torch.randint(0, 16, (16,), device=DEV)
    losses, _, _ = _train(m, x, y, 6, lr=0.05)
    for a, b in zip(r["losses"], losses):
        assert abs(a - b) < 0.05 * max(1.0, abs(b)), (r["losses"], losses)


@pytest.mark.parametrize("model_name,stages,n_micro,batch", [("mnist", 2, 4, 16), ("tiny_resnet", 2, 4, 32)])
def test_pipeline_stages_share_the_gpu_and_match_the_unsplit_cpu_model(tmp_path, model_name, stages, n_micro, batch):
    """The pipeline engine with every stage on GPU 0 (one process per stage; activations and gradients host-staged
    over gloo, the transport the reference's MP script uses): the HIP engine under a 1F1B schedule - several
    micro-batches in flight through the same fused blocks, gradients accumulated over micro-batches in the flat
    buffer - against the unsplit model on the CPU processing the same micro-batches."""
    from distributedpytorch_amd import launch
    from distributedpytorch_amd.parallel.pp import partition_balanced, param_costs
    from distributedpytorch_amd.utils import testing as T
    import test_pp_gloo as P
    url = f"tcp://127.0.0.1:{T.free_port()}"
    launch.spawn(T.pp_worker, [stages], 0, args=(url, model_name, str(tmp_path), stages, 1, n_micro, "1f1b", batch, 3, True))
    layers, ref_loss = P._reference(model_name, batch, n_micro)
    bounds = [(0, 1), (1, 2)] if model_name == "mnist" else partition_balanced(param_costs(layers), stages)
    final = torch.load(tmp_path / f"final_r0_s{stages - 1}.pt", weights_only=True)
    assert abs(final["losses"][0] - ref_loss) < (1e-4 if model_name == "mnist" else 0.05) * max(1.0, abs(ref_loss))
    assert final["losses"][-1] < final["losses"][0]
    num = den = 0.0
    for s in range(stages):
        a, b = bounds[s]
        got = torch.load(tmp_path / f"grads_r0_s{s}.pt", weights_only=True)
        for li in range(a, b):
            for n, p in layers[li].named_parameters():
                g = got[n if b - a == 1 else f"{li - a}.{n}"]
                if model_name == "mnist":                      # fp32 kernels: per-tensor, tight
                    assert torch.allclose(g, p.grad, rtol=2e-3, atol=2e-5), (s, n)
                if g.shape == p.grad.shape:                    # (the stem filter is channel-padded on the GPU)
                    num += float((g - p.grad).pow(2).sum()); den += float(p.grad.pow(2).sum())
    if model_name != "mnist":
        # bf16 activations, ReLU bits taken from rounded values: ONE process on the GPU is already 11-12 % (L2, all
        # parameters) from the fp32 CPU gradient on this model; a dropped or doubled micro-batch would be >= 25 %
        assert (num / den) ** 0.5 < 0.2, (num / den) ** 0.5
        # the sharp check: the unsplit model on the GPU over the same micro-batches
        from distributedpytorch_amd import ops
        from distributedpytorch_amd.models import resnet
        from distributedpytorch_amd.parallel.flat import FlatParams
        from distributedpytorch_amd.parallel.pp import split_microbatches
        torch.manual_seed(1234)
        full = resnet.tiny_resnet().train().to(DEV)
        FlatParams(full)
        glayers = full.pipeline_layers()
        x, y = T._data((3, 32, 32), 16, batch)
        x, y = x.to(DEV), y.to(DEV)
        off = 0
        for sz in split_microbatches(batch, n_micro):
            h = x[off:off + sz]
            for l in glayers:
                h = l(h)
            (ops.softmax_cross_entropy(h, y[off:off + sz], 16) * (sz / float(batch))).backward()
            off += sz
        torch.cuda.synchronize()
        num = den = 0.0
        for s in range(stages):
            a, b = bounds[s]
            got = torch.load(tmp_path / f"grads_r0_s{s}.pt", weights_only=True)
            for li in range(a, b):
                for n, p in glayers[li].named_parameters():
                    g = got[n if b - a == 1 else f"{li - a}.{n}"]
                    num += float((g - p.grad.cpu()).pow(2).sum()); den += float(p.grad.pow(2).sum())
        assert (num / den) ** 0.5 < 0.05, (num / den) ** 0.5


@pytest.mark.parametrize("workers", [1, 3])
def test_prefetch_loader_delivers_the_same_batches_on_the_device(workers):
    """Pinned staging + copy-stream H2D ahead of the consumer: same order, same bytes as the in-line loader, usable
    on the current stream right away (the consumer waits on the copy's event)."""
    from distributedpytorch_amd.data import BatchLoader, PrefetchLoader, SyntheticImageDataset
    ds = SyntheticImageDataset(70, (3, 32, 32), 10, seed=9)
    plain = BatchLoader(ds, 16, shuffle=True, seed=2)
    pre = PrefetchLoader(BatchLoader(ds, 16, shuffle=True, seed=2), torch.device(DEV, 0), depth=2, workers=workers)
    sums = []
    for (xa, ya), (xb, yb) in zip(plain, pre):
        assert xb.is_cuda and yb.is_cuda
        sums.append((xb * 2).sum())                       # consumed on the compute stream immediately
        assert torch.equal(xa, xb.cpu()) and torch.equal(ya, yb.cpu())
    assert len(sums) == 5


@pytest.mark.parametrize("model_name", ["mnist", "tiny_resnet"])
def test_sharded_optimizer_on_the_gpu_matches_the_replicated_one(tmp_path, model_name):
    """ZeRO stage 1 with both ranks on GPU 0: the fused SGD kernel over a shard of the flat buffers (shard-relative
    weight-decay table), reduce-to-owner per bucket piece, parameter broadcasts, bf16 shadows re-derived."""
    from distributedpytorch_amd import launch
    from distributedpytorch_amd.utils import testing as T
    a, b = tmp_path / "zero", tmp_path / "ref"
    a.mkdir(); b.mkdir()
    launch.spawn(T.dp_worker, [2], 0, args=(f"tcp://127.0.0.1:{T.free_port()}", model_name, str(a), 64, 4, "zero", "gloo_gpu"))
    launch.spawn(T.dp_worker, [2], 0, args=(f"tcp://127.0.0.1:{T.free_port()}", model_name, str(b), 64, 4, "zero_ref", "gloo_gpu"))
    z = [torch.load(a / f"final_{r}.pt", weights_only=True) for r in range(2)]
    ref = torch.load(b / "final_0.pt", weights_only=True)
    assert z[0]["momentum_numel"] + z[1]["momentum_numel"] == ref["momentum_numel"]
    num = den = 0.0
    for name, p in ref["params"].items():
        assert torch.equal(z[0]["params"][name], z[1]["params"][name]), name            # replicas bit-identical
        num += float((z[0]["params"][name] - p).pow(2).sum()); den += float(p.pow(2).sum())
    # mnist: fp32 kernels (atomics order only); tiny_resnet: bf16 + ReLU bits, compared in aggregate
    assert (num / den) ** 0.5 < (1e-4 if model_name == "mnist" else 2e-2), (num / den) ** 0.5
    assert z[0]["losses"][-1] < z[0]["losses"][0]


def test_two_ranks_on_the_gpu_reduce_through_the_native_layers_python_side(tmp_path, monkeypatch):
    """As test_two_ranks_share_the_gpu..., with utils/fake_native.GlooNative installed as ``Communicator.native``
    (DPA_FAKE_NATIVE=1): DEVICE buckets of the HIP engine go through ``_native_all_reduce`` / ``_NativeWork`` - the path
    of a multi-GPU job - after the native self-test's collective phases have run at world size 2."""
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    test_two_ranks_share_the_gpu_and_match_the_cpu_reference(tmp_path, "mnist")
    calls = torch.load(tmp_path / "final_1.pt", weights_only=True)["native_calls"]
    assert calls[:7] == ["all_reduce", "all_reduce", "broadcast", "reduce", "reduce_scatter", "all_gather", "batch_p2p"]
    assert calls.count("all_reduce") >= 2 + 3


def test_pipeline_stages_on_the_gpu_exchange_through_grouped_native_p2p(tmp_path, monkeypatch):
    """As test_pipeline_stages_share_the_gpu..., with the stand-in native layer: activations and gradients of the HIP
    engine leave as DEVICE tensors through ``Communicator.batch_p2p`` -> ``native.batch_p2p`` (the grouped ncclSend /
    ncclRecv form), on sub-communicators that build_grid made with ``native.split``."""
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    test_pipeline_stages_share_the_gpu_and_match_the_unsplit_cpu_model(tmp_path, "mnist", 2, 4, 16)
    calls = torch.load(tmp_path / "final_r0_s1.pt", weights_only=True)["native_calls"]
    assert calls["pp"].count("batch_p2p") >= 3 * 4
