"""The node-partitioned layers on the HIP backend (world size 1: the exchange degenerates to a copy) must equal
the ordinary single-GPU layers -- checks HipBackend, the rectangular segment reduce and its transpose, and the
autograd chain through all_gather_rows.  The cross-rank logic is covered on CPU by tests/test_partition_gloo.py.
Needs an MI355X."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_partitioned_gcn_and_rgcn_world1_match_fused_layers():
    from biomedkg_amd import ops, partition as P, synthetic
    from biomedkg_amd.graph import graph_of, rel_graph_of
    from biomedkg_amd.model.encoder import GCNEncoder, RGCN
    g = synthetic.primekg_like(3000, 60_000, 8, seed=2)
    ei, et, n = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"]
    torch.manual_seed(0)
    part = P.NodePartition(ei, n, 0, 1)
    assert part.bounds == [0, n] and part.n_max == n

    keep = torch.rand(ei.size(1), device=DEV) >= 0.4
    enc = GCNEncoder(64, 32, 32, 2).to(DEV).eval()
    x = torch.randn(n, 64, device=DEV)
    pg = P.PartitionedGraph(part, ei, P.HipBackend, keep=keep)
    out_p = P.gcn_encoder_partitioned(enc, x, pg, P.HipBackend)
    out_p.pow(2).sum().backward()
    grads_p = [p.grad.clone() for p in enc.parameters()]
    enc.zero_grad()
    out = enc(x, ei, edge_keep=keep)
    out.pow(2).sum().backward()
    torch.testing.assert_close(out_p, out, rtol=1e-4, atol=1e-5)
    for a, p in zip(grads_p, enc.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-3, atol=1e-4)

    renc = RGCN(64, 32, 32, 1, g["num_relations"]).to(DEV).eval()
    prg = P.PartitionedRelGraph(part, ei, et, g["num_relations"], P.HipBackend)
    xr = x.clone().requires_grad_(True)
    z_p = P.rgcn_encoder_partitioned(renc, xr, prg, P.HipBackend)
    z_p.pow(2).sum().backward()
    grads_p, dx_p = [p.grad.clone() for p in renc.parameters()], xr.grad.clone()
    renc.zero_grad()
    xr2 = x.clone().requires_grad_(True)
    z = renc(xr2, ei, et)
    z.pow(2).sum().backward()
    torch.testing.assert_close(z_p, z, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx_p, xr2.grad, rtol=1e-3, atol=1e-4)
    for a, p in zip(grads_p, renc.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-3, atol=1e-3)


def test_partitioned_training_losses_world1_match_the_modules():
    """``bench.py --mode partition`` at one rank: the partitioned GGD / KGE losses and every parameter gradient equal
    the whole-graph modules' (same corruption permutation, same negatives, dropout off)."""
    from types import SimpleNamespace
    from biomedkg_amd import partition as P, rand as R, sampling, synthetic
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(3000, 60_000, 8, seed=4)
    ei, et, n, r = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"], g["num_relations"]
    part = P.NodePartition(ei, n, 0, 1)
    torch.manual_seed(0)
    # --- GGD
    gcl = GGDModule(in_dim=64, hidden_dim=32, out_dim=32, num_hidden_layers=2, fuse_method="none").to(DEV).eval()
    x = torch.randn(n, 64, device=DEV)
    perm = torch.randperm(n, device=DEV)
    pg = P.PartitionedGraph(part, ei, P.HipBackend)
    loss_p = P.ggd_loss_partitioned(gcl.model, part.shard(x), pg, P.HipBackend, perm)
    loss_p.backward()
    grads_p = [p.grad.clone() for p in gcl.model.parameters()]
    gcl.model.zero_grad()
    pos_h = gcl.model.encoder(x, ei)
    neg_h = gcl.model.encoder(x[perm], ei)
    head = gcl.model.mlp[-1]
    pred = torch.cat([torch.nn.functional.linear(pos_h, head.weight, head.bias).sum(1),
                      torch.nn.functional.linear(neg_h, head.weight, head.bias).sum(1)])
    gt = torch.cat([torch.ones(n, device=DEV), torch.zeros(n, device=DEV)])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt)
    loss.backward()
    torch.testing.assert_close(loss_p, loss, rtol=1e-4, atol=1e-5)
    for a, p in zip(grads_p, gcl.model.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=2e-3, atol=1e-4)
    # --- KGE
    kge = KGEModule("rgcn", "dismult", 64, 32, 32, 2, r, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV).eval()
    neg = sampling.negative_sampling(ei, n, ei.size(1), seed=3)
    nt = et[torch.randperm(et.numel(), device=DEV)]
    prg = P.PartitionedRelGraph(part, ei, et, r, P.HipBackend)
    loss_p = P.kge_loss_partitioned(kge.model, part.shard(x), prg, P.HipBackend, ei, et, neg, nt, kge.model.decoder.kind)
    loss_p.backward()
    grads_p = [p.grad.clone() for p in kge.model.parameters()]
    kge.model.zero_grad()
    z = kge.model.encode(x, ei, et)
    pred = torch.cat([kge.model.decode(z, ei, et), kge.model.decode(z, neg, nt)])
    gt = torch.cat([torch.ones(ei.size(1), device=DEV), torch.zeros(neg.size(1), device=DEV)])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt) + 1e-2 * (
        z.pow(2).mean() + kge.model.decoder.rel_emb.pow(2).mean())
    loss.backward()
    torch.testing.assert_close(loss_p, loss, rtol=1e-4, atol=1e-5)
    for a, p in zip(grads_p, kge.model.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=2e-3, atol=1e-4)


def test_partitioned_step_world1_in_bf16_mode_runs_the_fused_views_path():
    """bf16 mode at a size that engages the side-by-side GGD views and the bf16 tables: the partitioned losses and
    gradients (exchange degenerate at one rank) against the whole-graph modules in the same mode, same permutation,
    same edge mask, same negatives."""
    from biomedkg_amd import ops, partition as P, rand as R, sampling, synthetic
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(6000, 120_000, 8, seed=6)
    ei, et, n, r = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"], g["num_relations"]
    part = P.NodePartition(ei, n, 0, 1)
    torch.manual_seed(0)
    x = torch.randn(n, 256, device=DEV)
    perm = torch.randperm(n, device=DEV)
    keep = torch.rand(ei.size(1), device=DEV) >= 0.4
    ops.set_matmul_precision("bf16")
    try:
        gcl = GGDModule(in_dim=256, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).eval()
        pg = P.PartitionedGraph(part, ei, P.HipBackend)
        calls = []
        orig = ops.gcn_conv_views
        ops.gcn_conv_views = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            loss_p = P.ggd_loss_partitioned(gcl.model, x, pg, P.HipBackend, perm, keep=keep)
        finally:
            ops.gcn_conv_views = orig
        assert len(calls) == 4                                           # the fused side-by-side path was taken
        loss_p.backward()
        grads_p = [p.grad.clone() for p in gcl.model.parameters()]
        gcl.model.zero_grad()

        class Fixed(R.DeviceRand):                                       # same draws for the module's own forward
            def coin(self): return 0.0
            def masked(self, xx, p): return xx
            def edge_keep(self, e, p, device): return keep
            def perm(self, nn, device): return perm
        with R.use(Fixed(seed=1)):
            pos_h, neg_h = gcl.model(x, ei)
        pred = torch.cat([pos_h, neg_h])
        gt = torch.cat([torch.ones(n, device=DEV), torch.zeros(n, device=DEV)])
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt)
        loss.backward()
        torch.testing.assert_close(loss_p, loss, rtol=1e-3, atol=1e-4)
        for a, p in zip(grads_p, gcl.model.parameters()):
            err = float((a - p.grad).norm() / p.grad.norm().clamp_min(1e-12))
            assert err < 2e-2, err
        # --- KGE
        kge = KGEModule("rgcn", "dismult", 256, 256, 256, 2, r, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV).eval()
        neg = sampling.negative_sampling(ei, n, ei.size(1), seed=3)
        nt = et[torch.randperm(et.numel(), device=DEV)]
        prg = P.PartitionedRelGraph(part, ei, et, r, P.HipBackend)
        loss_p = P.kge_loss_partitioned(kge.model, x, prg, P.HipBackend, ei, et, neg, nt, kge.model.decoder.kind)
        loss_p.backward()
        grads_p = [p.grad.clone() for p in kge.model.parameters()]
        kge.model.zero_grad()
        z = kge.model.encode(x, ei, et)
        ce, _, _ = ops.kge_bce_loss(z, kge.model.decoder.rel_emb, ei, et, neg, nt, kge.model.decoder.kind)
        loss = ce + 1e-2 * (z.pow(2).mean() + kge.model.decoder.rel_emb.pow(2).mean())
        loss.backward()
        torch.testing.assert_close(loss_p, loss, rtol=1e-3, atol=1e-4)
        for a, p in zip(grads_p, kge.model.parameters()):
            err = float((a - p.grad).norm() / p.grad.norm().clamp_min(1e-12))
            assert err < 2e-2, err
    finally:
        ops.set_matmul_precision("fp32")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_row_partitioned_ntxent_world1_matches_the_whole_problem(mode):
    """partition.ntxent_partitioned at one rank = the rectangular (anchor subset) kernels over ALL rows: same loss and
    gradients as the symmetric whole-problem path."""
    from biomedkg_amd import ops, partition as P, synthetic
    g = synthetic.primekg_like(1500, 20_000, 4, seed=1)
    part = P.NodePartition(g["edge_index"].to(DEV), g["num_nodes"], 0, 1)
    torch.manual_seed(2)
    a = torch.randn(1500, 96, device=DEV)
    b = 0.6 * a + 0.4 * torch.randn(1500, 96, device=DEV)
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ops.ntxent_loss(a1, b1, 0.2, operands=mode)
    (1.7 * ref).backward()
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    got = P.ntxent_partitioned(a2, b2, 0.2, part, ops.NtXentKernels(operands=mode))
    (1.7 * got).backward()
    tol = 1e-6 if mode == "fp32" else 1e-5
    assert abs(float(got) - float(ref)) <= tol * abs(float(ref)), (float(got), float(ref))
    for x, y in ((a2.grad, a1.grad), (b2.grad, b1.grad)):
        assert float((x - y).norm() / y.norm()) < (1e-5 if mode == "fp32" else 1e-3)


def _two_rank_worker(rank, world, port, outdir):
    """Two processes on ONE MI355X, gloo as the transport (CUDA tensors go through the host): functional check of
    the fused partitioned layers' exchange placement at world 2 -- never a performance statement."""
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from biomedkg_amd import ops, partition as P, sampling, synthetic
    from biomedkg_amd.gcl_module import GGDModule, GRACEModule
    from biomedkg_amd.kge_module import KGEModule
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops.set_matmul_precision("bf16")
    g = synthetic.primekg_like(12000, 240_000, 8, seed=6)
    ei, et, n, r = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"], g["num_relations"]
    part = P.NodePartition(ei, n, rank, world)
    torch.manual_seed(0)
    x = torch.randn(n, 256, device=DEV)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(DEV)
    keep = (torch.rand(ei.size(1), generator=torch.Generator().manual_seed(2)) >= 0.4).to(DEV)
    torch.manual_seed(5)
    gcl = GGDModule(in_dim=256, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).eval()
    kge = KGEModule("rgcn", "dismult", 256, 256, 256, 2, r, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV).eval()
    grace = GRACEModule(in_dim=256, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).eval()
    out = {}
    pg = P.PartitionedGraph(part, ei, P.HipBackend)
    loss = P.ggd_loss_partitioned(gcl.model, part.shard(x), pg, P.HipBackend, perm, keep=keep)
    loss.backward()
    grads = [p.grad.clone() for p in gcl.model.parameters()]
    for t in grads:
        dist.all_reduce(t); t /= world
    out["ggd"] = (P.loss_value(loss, world).cpu(), [t.cpu() for t in grads])
    neg = sampling.negative_sampling(ei, n, ei.size(1), seed=3)
    nt = et[torch.randperm(et.numel(), generator=torch.Generator().manual_seed(3)).to(DEV)]
    prg = P.PartitionedRelGraph(part, ei, et, r, P.HipBackend)
    sh = slice(rank, None, world)
    loss = P.kge_loss_partitioned(kge.model, part.shard(x), prg, P.HipBackend, ei[:, sh].contiguous(), et[sh].contiguous(),
                                  neg[:, sh].contiguous(), nt[sh].contiguous(), kge.model.decoder.kind,
                                  n_pos_global=ei.size(1), n_neg_global=neg.size(1))
    loss.backward()
    grads = [p.grad.clone() for p in kge.model.parameters()]
    for t in grads:
        dist.all_reduce(t); t /= world
    out["kge"] = (P.loss_value(loss, world).cpu(), [t.cpu() for t in grads])
    # GRACE head on the rank's rows: encoder over the shard (no augmentation), projector, row-partitioned NT-Xent
    z = P.gcn_encoder_partitioned(grace.model.encoder, part.shard(x), pg, P.HipBackend)
    h1 = grace.model.project(z)
    h2 = grace.model.project(z * 0.5 + 0.1)
    loss = P.ntxent_partitioned(h1, h2, 0.2, part)
    loss.backward()
    grads = [p.grad.clone() for p in grace.model.parameters()]
    for t in grads:
        dist.all_reduce(t); t /= world
    out["grace"] = (P.loss_value(loss, world).cpu(), [t.cpu() for t in grads])
    torch.save(out, os.path.join(outdir, f"w{world}_r{rank}.pt"))
    dist.destroy_process_group()


def test_fused_partitioned_steps_two_ranks_on_one_gpu_match_one_rank(tmp_path):
    import socket
    import torch.multiprocessing as mp
    res = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_two_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        res[world] = torch.load(tmp_path / f"w{world}_r0.pt")
    for key in ("ggd", "kge", "grace"):
        l1, g1 = res[1][key]
        l2, g2 = res[2][key]
        assert abs(float(l1) - float(l2)) <= 2e-3 * abs(float(l1)), (key, float(l1), float(l2))
        for a, b in zip(g1, g2):
            err = float((a - b).norm() / a.norm().clamp_min(1e-12))
            assert err < 3e-2, (key, err)


def test_collectives_of_the_partitioned_step_over_a_one_rank_rccl_group():
    """BKG_PARTITION_FORCE_COLLECTIVES=1: bench.py's node-partitioned step at ONE rank with every collective of the
    N-rank path issued for real over RCCL (bf16 all-gather of the gather tables, fp32 reduce-scatter of their gradients,
    the flat gradient all-reduce) -- the closest a single-GPU box gets to the N-rank RCCL path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BKG_PARTITION_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "partition", "--steps", "2",
                          "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["value"] > 0 and line["n_gpus"] == 1


def test_bench_starts_its_own_ranks():
    """``python3 bench.py --gpus 2`` with NO launcher around it (the driver's SCALE command form; the reference's users
    run one command too: train_gcl.py:63-68, Lightning starts the ranks): the parent spawns two fresh rank processes
    before anything touches the GPU, relays rank 0's single JSON line and returns the worst exit code.  Here the two
    ranks share the one GPU of the box over gloo (BKG_BENCH_ONE_DEVICE / BKG_BENCH_BACKEND); without those variables
    the same command selects RCCL with one device per rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BKG_BENCH_ONE_DEVICE="1", BKG_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"
    assert line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["parallelism"] == "np2"


def test_bench_ends_the_job_when_a_rank_dies():
    """A rank that dies after the rendezvous (BKG_BENCH_FAIL_RANK: fault injection) must not leave the launcher waiting on
    rank 0's collective until the caller's own limit: the other ranks are ended and a non-zero code comes back promptly."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BKG_BENCH_ONE_DEVICE="1", BKG_BENCH_BACKEND="gloo", BKG_BENCH_FAIL_RANK="1")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert time.time() - t0 < 300
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
