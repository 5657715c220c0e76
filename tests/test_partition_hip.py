"""The node-partitioned layers on the HIP backend (world size 1: the exchange degenerates to a copy) must equal
the ordinary single-GPU layers -- checks HipBackend, the rectangular segment reduce and its transpose, and the
autograd chain through all_gather_rows.  The cross-rank logic is covered on CPU by tests/test_partition_gloo.py.
Needs an MI355X."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("kept_views", [False, True])
def test_partitioned_gcn_and_rgcn_world1_match_fused_layers(monkeypatch, kept_views):
    """``kept_views``: the dropout_edge view as CSRs of its own on BOTH sides (graph.KEPT_CSR_MIN_EDGES = 0: the
    partitioned graph cuts them out of its weighted CSRs, the whole graph builds them from the mask)."""
    from biomedkg_amd import graph as G, ops, partition as P, synthetic
    from biomedkg_amd.graph import graph_of, rel_graph_of
    from biomedkg_amd.model.encoder import GCNEncoder, RGCN
    monkeypatch.setattr(G, "KEPT_CSR_MIN_EDGES", 0 if kept_views else 1 << 60)
    g = synthetic.primekg_like(3000, 60_000, 8, seed=2)
    ei, et, n = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"]
    torch.manual_seed(0)
    part = P.NodePartition(ei, n, 0, 1)
    assert part.bounds == [0, n] and part.n_max == n

    keep = torch.rand(ei.size(1), device=DEV) >= 0.4
    enc = GCNEncoder(64, 32, 32, 2).to(DEV).eval()
    x = torch.randn(n, 64, device=DEV)
    pg = P.PartitionedGraph(part, ei, P.HipBackend, keep=keep)
    assert isinstance(pg.fwd, G.KeptCSR) == kept_views and isinstance(pg.bwd, G.KeptCSR) == kept_views
    if kept_views:      # kept edges that are no self loops + one unit loop per node, on both sides
        live = int((keep & (ei[0] != ei[1])).sum()) + n
        assert int(pg.fwd.count) == live == int(pg.bwd.count)
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
    loss_p = P.kge_loss_partitioned(kge.model, part.shard(x), prg, P.HipBackend, P.TripletShare(part, ei, et),
                                    P.TripletShare(part, neg, nt), kge.model.decoder.kind,
                                    n_pos_global=ei.size(1), n_neg_global=neg.size(1))
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


@pytest.mark.parametrize("kept_views", [False, True])
def test_partitioned_step_world1_in_bf16_mode_runs_the_fused_views_path(monkeypatch, kept_views):
    """bf16 mode at a size that engages the side-by-side GGD views and the bf16 tables: the partitioned losses and
    gradients (exchange degenerate at one rank) against the whole-graph modules in the same mode, same permutation,
    same edge mask, same negatives.  ``kept_views``: the partitioned side propagates over the view's own CSRs, the module
    over the masked weights."""
    from biomedkg_amd import graph as G, ops, partition as P, rand as R, sampling, synthetic
    monkeypatch.setattr(G, "KEPT_CSR_MIN_EDGES", 1 << 60)
    # (the partitioned step keeps GGD's last conv layer as a layer; the whole-graph module it is compared with does too here:
    # its default -- that layer folded into the linear head -- rounds in other places, tests/test_hip_bf16_mode.py)
    monkeypatch.setattr(ops, "GGD_FOLD_LAST", False)
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
            G.KEPT_CSR_MIN_EDGES = 0 if kept_views else 1 << 60
            loss_p = P.ggd_loss_partitioned(gcl.model, x, pg, P.HipBackend, perm, keep=keep)
            assert isinstance(pg.gcn_weights(keep).dst, G.KeptCSR) == kept_views
        finally:
            ops.gcn_conv_views = orig
            G.KEPT_CSR_MIN_EDGES = 1 << 60
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
        loss_p = P.kge_loss_partitioned(kge.model, x, prg, P.HipBackend, P.TripletShare(part, ei, et),
                                        P.TripletShare(part, neg, nt), kge.model.decoder.kind,
                                        n_pos_global=ei.size(1), n_neg_global=neg.size(1))
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
    # both-owner scoring: every rank takes, from the SAME global triplet lists, the ones that touch its nodes
    loss = P.kge_loss_partitioned(kge.model, part.shard(x), prg, P.HipBackend, P.TripletShare(part, ei, et),
                                  P.TripletShare(part, neg, nt), kge.model.decoder.kind,
                                  n_pos_global=ei.size(1), n_neg_global=neg.size(1))
    loss.backward()
    grads = [p.grad.clone() for p in kge.model.parameters()]
    for t in grads:
        dist.all_reduce(t); t /= world
    out["kge"] = (P.loss_value(loss, world).cpu(), [t.cpu() for t in grads])
    # the DPI fine-tune's decoder (BASELINE configs[4]: TransE, every relation id forced to 1) over the same partition
    torch.manual_seed(6)
    dpi = KGEModule("rgcn", "transe", 256, 256, 256, 2, r, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV).eval()
    one = torch.full_like(et, 1)
    prg1 = P.PartitionedRelGraph(part, ei, one, r, P.HipBackend)
    loss = P.kge_loss_partitioned(dpi.model, part.shard(x), prg1, P.HipBackend, P.TripletShare(part, ei, one),
                                  P.TripletShare(part, neg, torch.full_like(nt, 1)), dpi.model.decoder.kind,
                                  n_pos_global=ei.size(1), n_neg_global=neg.size(1))
    loss.backward()
    grads = [p.grad.clone() for p in dpi.model.parameters()]
    for t in grads:
        dist.all_reduce(t); t /= world
    out["dpi_transe"] = (P.loss_value(loss, world).cpu(), [t.cpu() for t in grads])
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
    for key in ("ggd", "kge", "dpi_transe", "grace"):
        l1, g1 = res[1][key]
        l2, g2 = res[2][key]
        assert abs(float(l1) - float(l2)) <= 2e-3 * abs(float(l1)), (key, float(l1), float(l2))
        for a, b in zip(g1, g2):
            if float(a.norm()) == 0.0 and float(b.norm()) == 0.0:
                continue                                 # (relation rows no triplet uses under fix_edge_id)
            err = float((a - b).norm() / a.norm().clamp_min(1e-12))
            assert err < (3e-2 if key != "dpi_transe" else 6e-2), (key, err)   # TransE: a sum of signs (ties flip)


@pytest.mark.parametrize("mode", ["partition", "replicas"])
def test_collectives_of_the_partitioned_step_over_a_one_rank_rccl_group(mode):
    """BKG_PARTITION_FORCE_COLLECTIVES=1: bench.py's N-rank steps at ONE rank with every collective issued for real over
    RCCL -- the node-partitioned step (bf16 gathers of the tables and output gradients, the flat gradient all-reduce) and
    the replicas step (the default over N GPUs since round 5: the two flat gradient all-reduces, the KGE one started after
    its backward and finished under the next GGD sub-step) -- the closest a single-GPU box gets to the N-rank RCCL path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BKG_PARTITION_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--mode", mode, "--steps", "2",
                          "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == ("strong" if mode == "partition" else "weak") and line["value"] > 0 and line["n_gpus"] == 1
    assert line["backend"] == "rccl" and line["mode"] == mode


@pytest.mark.parametrize("keep_outer_ranks,mode", [(False, "partition"), (True, "partition"), (False, "auto")])
def test_bench_under_torch_distributed_run(keep_outer_ranks, mode):
    """The other form the driver may use: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...``.  The outer launcher's rank 0 becomes bench.py's own launcher and the
    other worker leaves: ``--mode partition`` = the node-partitioned step (graphed sub-steps first over RCCL, eager
    fallback; here: gloo on the one GPU); ``--mode auto`` (round 6) = BOTH regimes on fresh ranks, one line: headline the
    partition, ``replicas`` (one graph per rank, weak scaling) beside it.  ``BKG_BENCH_OUTER_RANKS=1`` keeps the outer
    launcher's processes as the ranks.  Every way: exit code 0, ONE JSON line."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BKG_BENCH_ONE_DEVICE="1", BKG_BENCH_BACKEND="gloo")
    if keep_outer_ranks:
        env["BKG_BENCH_OUTER_RANKS"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000",
                          "--mode", mode],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo" and line["value"] > 0
    assert line["scaling"] == "strong" and line["exchange"] is not None and line["mode"] == "partition"
    if mode == "auto":
        rep = line["replicas"]
        assert rep["scaling"] == "weak" and rep["mode"] == "replicas" and rep["parallelism"] == "dp2" and rep["value"] > 0
        ex = rep["exchange"]                 # the regime's only wire traffic: the two flat gradient buckets
        assert ex["exchanges_per_step"] == 2 and ex["bucket_bytes_kge"] > ex["bucket_bytes_ggd"] > 0
        assert ex["allreduce_alone_ms_kge"] > 0 and ex["wire_bytes_sent_per_step"] == ex["bucket_bytes_ggd"] + ex["bucket_bytes_kge"]
        assert [a["ok"] for a in line["attempts"]] == [True]


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
    lines = {}
    for mode in ("auto", "partition", "replicas"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                              "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000", "--mode", mode],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        got = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(got) == 1, out.stdout[-2000:]
        lines[mode] = json.loads(got[0])
        assert lines[mode]["n_gpus"] == 2 and lines[mode]["ranks"] == 2 and lines[mode]["backend"] == "gloo"
        assert lines[mode]["value"] > 0
    # the default over N GPUs (round 6): both regimes on one line -- headline the node partition (BASELINE configs[3], strong
    # scaling), beside it the replicas (one graph per rank, gradient all-reduce only: the reference's DDP regime, weak)
    assert lines["auto"]["mode"] == "partition" and lines["auto"]["scaling"] == "strong"
    assert lines["auto"]["config"]["parallelism"] == "np2"
    rep = lines["auto"]["replicas"]
    assert rep["mode"] == "replicas" and rep["scaling"] == "weak" and rep["parallelism"] == "dp2" and rep["value"] > 0
    assert lines["partition"]["scaling"] == "strong" and lines["partition"]["config"]["parallelism"] == "np2"
    assert lines["replicas"]["scaling"] == "weak" and lines["replicas"]["config"]["parallelism"] == "dp2"
    assert "replicas" not in lines["partition"] and "replicas" not in lines["replicas"]


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


# ------------------------------------------------------------------------------------------- round 4: both-owner scoring
def _slot_table(part, full):
    """[N, d] -> the [world * n_max, d] table in slot space (what an all-gather of the shards produces)."""
    table = torch.zeros(part.world * part.n_max, full.size(1), dtype=full.dtype, device=full.device)
    table[part.node_slots()] = full
    return table


@pytest.mark.parametrize("world", [1, 3, 8])
def test_block_negative_sampler_is_pyg_negative_sampling_cut_by_owner_blocks(world):
    """sampling.BlockNegativeSampler (csrc/negsample.hip bkg_negative_sample_blocks): over all ranks the counted (head-owned)
    negatives are K distinct off-diagonal non-edges -- PyG negative_sampling's contract, reference kge_module.py:93 --
    allotted to the (head owner, tail owner) blocks in proportion to their free cells; the two owners of a block draw the
    SAME triplets and relation ids; the ids over all K positions are edge_type.repeat(k)[a permutation] (kge_module.py:97-
    100) -- dealt to the blocks once, relation-grouped inside a block, every relation in every block in proportion; the
    incidence positions handed to the scorer are exactly the entries with a local node; another seed gives other
    negatives."""
    from biomedkg_amd import partition as P, sampling, synthetic
    g = synthetic.primekg_like(3000, 60_000, 8, seed=2)
    ei, et, n = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"]
    ratio = 2
    k = ratio * ei.size(1)
    pos_codes = set((ei[0] * n + ei[1]).tolist())
    blocks, counted = {}, []
    for rank in range(world):
        part = P.NodePartition(ei, n, rank, world)
        smp = sampling.BlockNegativeSampler(ei, et, part.bounds, rank, k)
        idx, types, n_own = smp.draw(seed=77), smp.types, smp.n_own
        assert smp.shortfall() == 0
        assert torch.equal(torch.cat([smp.types_own, smp.types_foreign]), types)
        # the ids of the two ranges come with their relation order (computed once, cached on the tensors)
        from biomedkg_amd import ops
        for t in (smp.types_own, smp.types_foreign):
            if t.numel():
                order, ptr = ops.relation_order_of(t, smp.num_relations)
                assert bool((t[order.long()][1:] >= t[order.long()][:-1]).all()) and int(ptr[-1]) == t.numel()
        # with n_max the same draw comes out in table slots
        slot_smp = sampling.BlockNegativeSampler(ei, et, part.bounds, rank, k, n_max=part.n_max)
        assert torch.equal(slot_smp.draw(seed=77), part.slots(idx))
        assert idx.shape == (2, smp.k_rank) and types.shape == (smp.k_rank,) and n_own == smp.n_own
        h_loc, t_loc = part.is_local(idx[0]), part.is_local(idx[1])
        assert bool(h_loc[:n_own].all()) and not bool(h_loc[n_own:].any()) and bool(t_loc[n_own:].all())
        # the incidences with a local node, as the sampler lists them by construction
        flat = idx.reshape(-1)
        want = torch.nonzero(part.is_local(flat)).squeeze(1)
        assert torch.equal(torch.sort(smp.entries).values, want)
        bt = torch.tensor(part.bounds[1:-1], device=DEV)
        for bid, out0, kq in smp.layout:
            a, b = divmod(bid, world)
            h, t = idx[0, out0:out0 + kq], idx[1, out0:out0 + kq]
            assert bool((torch.bucketize(h, bt, right=True) == a).all()) and bool((torch.bucketize(t, bt, right=True) == b).all())
            triple = torch.stack([h, t, types[out0:out0 + kq]]).cpu()
            if bid in blocks:
                assert torch.equal(blocks[bid], triple), f"the two owners of block {bid} drew different triplets"
            blocks[bid] = triple
        counted.append(torch.cat([idx[:, :n_own], types[None, :n_own]]).cpu())
        assert torch.equal(smp.draw(seed=77), idx)                                    # a pure function of (graph, seed)
        assert not torch.equal(smp.draw(seed=78), idx)
    assert len(blocks) == world * world
    allneg = torch.cat(counted, dim=1)
    assert allneg.size(1) == k
    codes = (allneg[0] * n + allneg[1]).tolist()
    assert len(set(codes)) == k and not (set(codes) & pos_codes)
    assert bool((allneg[0] != allneg[1]).all())
    # relation ids: a permutation of edge_type.repeat(ratio) over the K global positions
    assert torch.equal(torch.sort(allneg[2]).values, torch.sort(et.repeat(ratio)).values.cpu())
    # quotas follow the blocks' free cells (proportional allocation: within one of the exact share)
    part = P.NodePartition(ei, n, 0, world)
    sizes = [b - a for a, b in zip(part.bounds[:-1], part.bounds[1:])]
    bt = torch.tensor(part.bounds[1:-1], device=DEV)
    loop_free = ei[:, ei[0] != ei[1]]
    cells = torch.unique(loop_free[0] * n + loop_free[1])
    used = torch.bincount(torch.bucketize(cells // n, bt, right=True) * world + torch.bucketize(cells % n, bt, right=True),
                          minlength=world * world).tolist()
    free = [sizes[a] * (sizes[b] - (a == b)) - used[a * world + b] for a in range(world) for b in range(world)]
    share = torch.bincount(et, minlength=8).double().cpu() / et.numel()
    for bid in range(world * world):
        assert abs(blocks[bid].size(1) - k * free[bid] / sum(free)) <= 1.0
        kb = blocks[bid].size(1)
        if kb >= 2000:                      # every relation turns up in every block in proportion (low-discrepancy deal)
            got = torch.bincount(blocks[bid][2], minlength=8).double() / kb
            assert float((got - share).abs().max()) < 0.02, (bid, got, share)


def test_block_negative_sampler_in_a_block_that_is_mostly_edges():
    """A dense block (a 60-node clique region: 70 % of its cells are positives) next to sparse ones: its quota follows its
    FREE cells, its candidates cover the shortfall risk (here: the whole block), nothing is a positive, nothing repeats."""
    from biomedkg_amd import partition as P, sampling
    gen = torch.Generator().manual_seed(0)
    n = 600
    dense = torch.nonzero(torch.rand(60, 60, generator=gen) < 0.7).t()
    sparse = torch.randint(0, n, (2, 3000), generator=gen)
    ei = torch.cat([dense, sparse], dim=1).to(DEV)
    et = torch.randint(0, 3, (ei.size(1),), generator=gen).to(DEV)
    pos_codes = set((ei[0] * n + ei[1]).tolist())
    k = ei.size(1)
    seen = []
    for rank in range(4):
        part = P.NodePartition(ei, n, rank, 4)
        smp = sampling.BlockNegativeSampler(ei, et, part.bounds, rank, k)
        idx, n_own = smp.draw(seed=5), smp.n_own
        assert smp.shortfall() == 0
        seen.append(idx[:, :n_own].cpu())
    allneg = torch.cat(seen, dim=1)
    codes = (allneg[0] * n + allneg[1]).tolist()
    assert allneg.size(1) == k and len(set(codes)) == k and not (set(codes) & pos_codes)
    assert bool((allneg[0] != allneg[1]).all())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("decoder", ["dismult", "transe"])
@pytest.mark.parametrize("size", ["small", "large"])
def test_both_owner_scoring_equals_the_whole_graph_scorer(mode, decoder, size):
    """ops.kge_bce_loss_owned over the shards of 1, 3 and 5 simulated ranks (the all-gather replaced by the table it would
    deliver) against ops.kge_bce_loss on the whole graph: the counted losses add up to the whole loss, every rank's dz is
    the whole dz of its rows (from its own nodes' incidence lists -- nothing is summed across ranks), the d rel_emb add
    up.  "large": shards above 4,096 rows (long-row tables, the stream form for DistMult); "small": the one-wave-per-row
    kernels."""
    from biomedkg_amd import ops, partition as P, sampling, synthetic
    n, e = (24_000, 400_000) if size == "large" else (1200, 15_000)
    g = synthetic.primekg_like(n, e, 8, seed=9)
    ei, et, r = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_relations"]
    kind = ops.KGE_DISTMULT if decoder == "dismult" else ops.KGE_TRANSE
    d = 256
    torch.manual_seed(3)
    z = torch.randn(n, d, device=DEV) * 0.3
    rel = torch.randn(r, d, device=DEV) * 0.3
    neg = sampling.negative_sampling(ei, n, ei.size(1), seed=11)
    nt = et[torch.randperm(et.numel(), device=DEV)]
    ops.set_matmul_precision(mode)
    try:
        zw, rw = z.clone().requires_grad_(True), rel.clone().requires_grad_(True)
        whole, _, _ = ops.kge_bce_loss(zw, rw, ei, et, neg, nt, kind, reg=1e-2)
        whole.backward()
        for world in ((1, 3) if size == "large" else (1, 3, 5)):
            losses, drel = [], torch.zeros_like(rel)
            dz = torch.zeros_like(z)
            for rank in range(world):
                part = P.NodePartition(ei, n, rank, world)
                zfull = z
                part.comm.all_gather = lambda shard, part=part: _slot_table(part, zfull.to(shard.dtype))
                pos_s, neg_s = P.TripletShare(part, ei, et), P.TripletShare(part, neg, nt)
                zl = part.shard(z).clone().requires_grad_(True)
                rl = rel.clone().requires_grad_(True)
                total = (ei.size(1) + neg.size(1)) / world
                loss = P.kge_bce_partitioned(zl, rl, pos_s, neg_s, kind, part, total, 1e-2, P.HipBackend)
                loss.backward()
                losses.append(float(loss))
                dz[part.lo:part.hi] = zl.grad / world
                drel += rl.grad / world
            tol = 1e-5 if mode == "fp32" else 2e-4
            assert abs(sum(losses) / world - float(whole)) <= tol * abs(float(whole)), (world, losses, float(whole))
            err_z = float((dz - zw.grad).norm() / zw.grad.norm())
            err_r = float((drel - rw.grad).norm() / rw.grad.norm())
            # TransE's gradient is a sum of signs: an element within rounding of zero flips one (as in the single-GPU tests)
            lim = (1e-5 if mode == "fp32" else 2e-3) if decoder == "dismult" else 5e-3
            assert err_z < lim and err_r < lim, (world, err_z, err_r)
    finally:
        ops.set_matmul_precision("fp32")


def _small_shard_worker(rank, world, port, outdir):
    """bf16 mode, halo exchange forced, shards of a few hundred rows: every row chunk of the first layer's fp32 product is
    below the streaming kernel's floor (this raised TypeError inside _exchanged_product)."""
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from biomedkg_amd import ops, partition as P, synthetic
    from biomedkg_amd.model.encoder import GCNEncoder
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops.set_matmul_precision("bf16")
    g = synthetic.primekg_like(1400, 20_000, 4, seed=8)
    ei, n = g["edge_index"].to(DEV), g["num_nodes"]
    part = P.NodePartition(ei, n, rank, world, halo=True)
    assert 256 <= part.n_local < 1024 and len(part.comm.chunks("in")) == 4
    torch.manual_seed(0)
    x = torch.randn(n, 256, device=DEV)
    enc = GCNEncoder(256, 256, 256, 1).to(DEV).eval()
    pg = P.PartitionedGraph(part, ei, P.HipBackend)
    out = P.gcn_encoder_partitioned(enc, part.shard(x).contiguous(), pg, P.HipBackend)
    out.float().pow(2).sum().backward()
    grads = [p.grad.clone() for p in enc.parameters()]
    for t in grads:
        dist.all_reduce(t)
    torch.save({"out": out.float().cpu(), "grads": [t.cpu() for t in grads], "lo": part.lo, "hi": part.hi},
               os.path.join(outdir, f"s{world}_r{rank}.pt"))
    dist.destroy_process_group()


def test_small_shards_with_chunked_halo_exchange_in_bf16_mode(tmp_path):
    import socket
    import torch.multiprocessing as mp
    from biomedkg_amd import ops, synthetic
    from biomedkg_amd.model.encoder import GCNEncoder
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_small_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [torch.load(tmp_path / f"s2_r{i}.pt") for i in range(2)]
    g = synthetic.primekg_like(1400, 20_000, 4, seed=8)
    ei, n = g["edge_index"].to(DEV), g["num_nodes"]
    ops.set_matmul_precision("bf16")
    try:
        torch.manual_seed(0)
        x = torch.randn(n, 256, device=DEV)
        enc = GCNEncoder(256, 256, 256, 1).to(DEV).eval()
        ref = enc(x, ei)
        ref.float().pow(2).sum().backward()
        got = torch.cat([p["out"] for p in parts]).to(DEV)
        assert float((got - ref.float()).norm() / ref.float().norm()) < 2e-2
        for a, p in zip(parts[0]["grads"], enc.parameters()):
            assert float((a.to(DEV) - p.grad).norm() / p.grad.norm().clamp_min(1e-12)) < 3e-2
    finally:
        ops.set_matmul_precision("fp32")


@pytest.mark.parametrize("ranks", [4, 8])
def test_bench_partitioned_on_four_and_eight_ranks_of_one_device(ranks):
    """``python3 bench.py --gpus 4`` / ``--gpus 8`` exactly as the driver's SCALE run issues it, with the ranks sharing
    this box's one GPU over gloo (BKG_BENCH_ONE_DEVICE / BKG_BENCH_BACKEND): skewed load-balanced shards, row chunks with
    empty per-peer lists, uneven and zero all-to-all counts, both-owner scoring with the block sampler -- and the exact-
    fp32 check losses of the N-rank run equal the one-rank run's (same graph, same weights, same draws)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BKG_BENCH_ONE_DEVICE="1", BKG_BENCH_BACKEND="gloo")
    common = ["--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000"]
    lines = {}
    for n in (1, ranks):
        # the N-rank run cuts its dropout_edge views' own CSRs (a 300 k-edge graph stays below the product's floor), the
        # one-rank run keeps the masked weights: the check losses must not tell
        run_env = dict(env, BKG_KEPT_CSR_MIN_EDGES="0") if n > 1 else env
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--mode", "partition"] + common,
                             env=run_env, capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, out.stderr[-3000:]
        got = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(got) == 1, out.stdout[-2000:]
        lines[n] = json.loads(got[0])
    line = lines[ranks]
    assert line["n_gpus"] == ranks and line["ranks"] == ranks and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["parallelism"] == f"np{ranks}"
    ex = line["exchange"]
    assert ex is not None and ex["wire_bytes_received_per_step"] > 0 and ex["exchanges_per_step"] > 0
    assert len(ex["rows"]["bounds"]) == ranks + 1
    for key in ("ggd", "kge"):
        a, b = lines[1]["check_loss"][key], line["check_loss"][key]
        assert abs(a - b) <= 1e-5 * abs(a), (key, a, b)


def test_block_sampler_is_rekeyed_by_the_replay_counter():
    """A captured step freezes the sampler's seed argument: the library's device replay counter (bkg_set_rng_epoch) re-keys
    the draw, so every replay gets fresh negatives -- and ranks that replay in lockstep (the same counter value) still draw
    the same blocks."""
    from biomedkg_amd import _lib as L, partition as P, sampling, synthetic
    g = synthetic.primekg_like(2000, 30_000, 5, seed=1)
    ei, et, n = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"]
    part = P.NodePartition(ei, n, 1, 3)
    smp = sampling.BlockNegativeSampler(ei, et, part.bounds, 1, ei.size(1))
    plain = smp.draw(seed=5)
    epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.call("bkg_set_rng_epoch", L.ptr(epoch))
    try:
        e0 = smp.draw(seed=5)
        epoch.add_(1)
        e1 = smp.draw(seed=5)
        e1_again = smp.draw(seed=5)
        other_rank = sampling.BlockNegativeSampler(ei, et, part.bounds, 0, ei.size(1))
        o1 = other_rank.draw(seed=5)
    finally:
        L.call("bkg_set_rng_epoch", None)
    assert torch.equal(e0, plain)                       # counter 0: the un-keyed draw
    assert not torch.equal(e1, e0) and torch.equal(e1, e1_again)
    # block (0, 1): rank 0 owns its heads, rank 1 its tails -- the same triplets at the same counter value
    b01 = 0 * 3 + 1
    lo_a, k_a = [(o, k) for b, o, k in other_rank.layout if b == b01][0]
    lo_b, k_b = [(o, k) for b, o, k in smp.layout if b == b01][0]
    assert k_a == k_b and torch.equal(o1[:, lo_a:lo_a + k_a], e1[:, lo_b:lo_b + k_b])


def test_partitioned_sub_steps_replay_as_hipgraphs_over_rccl():
    """BKG_BENCH_PARTITION_GRAPH=1 on a one-rank RCCL group with every collective of the N-rank path issued for real
    (BKG_PARTITION_FORCE_COLLECTIVES=1): each sub-step -- exchanges, gradient all-reduce, clip, Adam -- is captured once
    per host-side branch and replayed; the host's time per step drops several-fold; the run trains like the eager one
    (the fixed-draw check losses fall, and by about as much)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for mode in ("0", "1", "1-inline"):
        env = dict(os.environ, BKG_PARTITION_FORCE_COLLECTIVES="1", BKG_BENCH_PARTITION_GRAPH=mode[0], MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29541", BKG_BENCH_DEFER_UPDATE="0" if mode.endswith("inline") else "1")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "partition", "--steps",
                              "12", "--warmup", "3", "--no-extras", "--no-cpu-baseline", "--nodes", "20000", "--edges", "300000"],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        lines[mode] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    eager, graphed = lines["0"], lines["1"]
    gs = graphed["exchange"]["graphed_sub_steps"]
    assert eager["exchange"]["graphed_sub_steps"] is None
    assert gs["refused"] is None and gs["captured"] == 3 and gs["replayed"] >= 20        # GGD with / without augmentation, KGE
    # the KGE sub-step's gradient bucket is reduced on a stream and a communicator of its own under the next GGD sub-step,
    # its update lands at the top of the next KGE call: one deferred update per KGE step of the graphed form
    assert gs.get("deferred_updates", 0) >= 12
    assert graphed["exchange"]["host_enqueue_ms_per_step_graphed"] < 0.5 * eager["exchange"]["host_enqueue_ms_per_step"]
    inline = lines["1-inline"]            # the graphed form with the bucket reduced inside the sub-step's own graph
    assert inline["exchange"]["graphed_sub_steps"].get("deferred_updates", 0) == 0
    for line in (eager, graphed, inline):
        for key in ("ggd", "kge"):
            assert line["check_loss_after"][key] < line["check_loss"][key], (key, line["check_loss"], line["check_loss_after"])
    # deferred or inline: the same updates in the same order on the same draws (the replay counter keys them)
    for key in ("ggd", "kge"):
        a, b = graphed["check_loss_after"][key], inline["check_loss_after"][key]
        assert abs(a - b) <= 2e-2 * abs(b) + 1e-4, (key, a, b)
    for key in ("ggd", "kge"):          # the same number of optimizer steps from the same weights, other random draws
        drop_e = eager["check_loss"][key] - eager["check_loss_after"][key]
        drop_g = graphed["check_loss"][key] - graphed["check_loss_after"][key]
        assert abs(drop_e - drop_g) < 0.35 * abs(drop_e) + 1e-3, (key, drop_e, drop_g)
