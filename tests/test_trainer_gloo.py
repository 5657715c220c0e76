"""world_size-2 gloo run (CPU) of the Trainer stand-in's data-parallel bookkeeping: per-rank batches (set_rank /
per-rank seeds), metrics averaged and should_stop OR-ed over the ranks before the callbacks act, rank 0's checkpoint
directory and best path on every rank (the behaviours Lightning DDP gives the reference's train_*.py)."""
import os
import socket
import time

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Loader:
    def __init__(self, n):
        self.n, self.rank, self.world = n, 0, 1

    def set_rank(self, rank, world):
        self.rank, self.world = rank, world

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield torch.full((4, 3), float(i * self.world + self.rank))


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), CUDA_VISIBLE_DEVICES="")
    from biomedkg_amd import trainer as T
    from biomedkg_amd.compat import LightningModule

    class Toy(LightningModule):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 1)
            self.seen, self.draws, self.val_calls = [], [], 0

        def training_step(self, batch):
            self.seen.append(float(batch[0, 0]))
            self.draws.append(float(torch.rand(1)))
            return self.lin(batch).pow(2).mean()

        def validation_step(self, batch, i):
            self.val_calls += 1
            # rank 0 alone would keep improving (never stop), rank 1 alone would stop at once: the mean decides for both
            score = (1.0 / self.val_calls) if self.trainer.rank == 0 else (10.0 * self.val_calls)
            self.log("val_loss", torch.tensor(score))

        def test_step(self, batch, i):
            self.log("test_loss", self.lin(batch).pow(2).mean())

        def configure_optimizers(self):
            return {"optimizer": torch.optim.SGD(self.lin.parameters(), lr=1e-3)}

    T.seed_everything(5)
    torch.manual_seed(100 + rank)                                     # different init per rank: fit must broadcast rank 0's
    model = Toy()
    ckpt_dir = os.path.join(out, f"ckpt_{rank}_{time.time()}")        # what train_kge.py does: a per-rank time stamp
    cb = T.ModelCheckpoint(dirpath=ckpt_dir, monitor="val_loss", save_top_k=1, mode="min")
    es = T.EarlyStopping(monitor="val_loss", patience=2)
    tr = T.Trainer(max_epochs=10, check_val_every_n_epoch=1, gradient_clip_val=1.0, callbacks=[cb, es],
                   default_root_dir=ckpt_dir)
    tr.fit(model, _Loader(3), _Loader(1))
    res = tr.test(model, _Loader(1), ckpt_path="best")
    torch.save({"seen": model.seen, "draws": model.draws, "epochs": tr.current_epoch + 1, "dir": cb.dirpath,
                "best": cb.best_model_path, "w": model.lin.weight.detach().clone(), "test": res[0],
                "hist": [h.get("val_loss") for h in tr.history]}, os.path.join(out, f"t{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_trainer_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"t{i}.pt")) for i in range(2))
    assert r0["seen"][:3] == [0.0, 2.0, 4.0] and r1["seen"][:3] == [1.0, 3.0, 5.0]   # set_rank sharded the batches
    assert r0["draws"] != r1["draws"]                                                  # per-rank random streams
    assert r0["epochs"] == r1["epochs"] < 10                                           # both stopped, together
    assert r0["hist"] == r1["hist"]                                                    # the averaged metric
    assert r0["dir"] == r1["dir"] and r0["best"] == r1["best"] and os.path.exists(r0["best"])
    assert torch.equal(r0["w"], r1["w"])                                               # same best checkpoint loaded
    assert r0["test"] == r1["test"]
