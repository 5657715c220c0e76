"""world_size-2 gloo runs (CPU) of the multi-GPU glue: flat-bucket gradient averaging and parameter
broadcast give every rank the mean gradient / rank-0 weights, as Lightning DDP would."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from biomedkg_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)                      # different init and data per rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3, bias=False))
    D.broadcast_parameters(model, src=0)
    x = torch.randn(8, 6)
    model(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    D.GradBucket(model.parameters()).all_reduce()
    t = D.max_over_ranks(float(rank + 1), "cpu")
    torch.save({"w": [p.detach().clone() for p in model.parameters()], "local": local,
                "avg": [p.grad.clone() for p in model.parameters()], "tmax": t}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_grad_bucket_and_broadcast_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(2))
    for a, b in zip(r0["w"], r1["w"]):
        assert torch.equal(a, b)                                       # broadcast from rank 0
    for l0, l1, a0, a1 in zip(r0["local"], r1["local"], r0["avg"], r1["avg"]):
        torch.testing.assert_close(a0, (l0 + l1) / 2)                  # mean over ranks
        assert torch.equal(a0, a1)
    assert r0["tmax"] == r1["tmax"] == 2.0


def _worker_overlap(rank, world, port, out):
    """bench.py's overlapped schedule: model B's bucket is started after its backward and finished only after model
    A's whole sub-step (its own bucket included) -- the collectives of the two buckets interleave."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from biomedkg_amd import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(7)
    a, b = torch.nn.Linear(4, 3), torch.nn.Linear(5, 2)               # identical init on both ranks
    ba, bb = D.GradBucket(a.parameters()), D.GradBucket(b.parameters())
    torch.manual_seed(200 + rank)                                     # different data per rank
    xa, xb = torch.randn(6, 4), torch.randn(6, 5)
    b(xb).pow(2).sum().backward()
    local_b = [p.grad.clone() for p in b.parameters()]
    bb.begin()                                                        # in flight ...
    a(xa).pow(2).sum().backward()                                     # ... while the other model computes
    local_a = [p.grad.clone() for p in a.parameters()]
    ba.all_reduce()                                                   # ... and runs its own collective
    bb.finish()
    bb.finish()                                                       # idempotent
    torch.save({"la": local_a, "lb": local_b, "aa": [p.grad.clone() for p in a.parameters()],
                "ab": [p.grad.clone() for p in b.parameters()]}, os.path.join(out, f"o{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_overlapped_buckets_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker_overlap, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"o{i}.pt")) for i in range(2))
    for key_l, key_a in (("la", "aa"), ("lb", "ab")):
        for l0, l1, a0, a1 in zip(r0[key_l], r1[key_l], r0[key_a], r1[key_a]):
            torch.testing.assert_close(a0, (l0 + l1) / 2)
            assert torch.equal(a0, a1)
