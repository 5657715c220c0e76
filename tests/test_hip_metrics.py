"""Evaluation metrics of the KGE module on the device (biomedkg_amd/utils/metrics.py; reference kge_module.py:63-74,
134-206) against the CPU fp64 restatement of torchmetrics' binary definitions (oracle/metrics.py, itself pinned against
scikit-learn in tests/test_oracle_known_answers.py): AUROC / AveragePrecision / F1 on logits with exact ties and values
outside [0, 1]; the 10-bootstrap wrapper with the Poisson multiplicities injected into both sides; and
``validation_step`` + ``on_validation_epoch_end`` end to end on a golden KGE case whose scores were recorded from the
reference's own module.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scores(n, pos_rate, ties, seed, logits=True):
    rng = np.random.default_rng(seed)
    t = (rng.random(n) < pos_rate).astype(np.int64)
    s = rng.normal(size=n) * 2.0 + t * 1.2
    if ties:
        s = np.round(s * 2.0) / 2.0
    if not logits:
        s = 1.0 / (1.0 + np.exp(-s))
    return s.astype(np.float32), t


@pytest.mark.parametrize("n,pos_rate,ties,logits", [(5000, 0.5, False, True), (20000, 0.08, True, True),
                                                    (3000, 0.7, True, False), (64, 0.5, True, True)])
def test_binary_metrics_match_the_torchmetrics_definitions(n, pos_rate, ties, logits):
    from biomedkg_amd.utils import metrics as M
    from oracle import metrics as O
    s, t = _scores(n, pos_rate, ties, n, logits)
    sd, td = torch.from_numpy(s).to(DEV), torch.from_numpy(t).to(DEV).to(torch.int32)
    for cls, ref in ((M.AUROC, O.binary_auroc), (M.AveragePrecision, O.binary_average_precision), (M.F1Score, O.binary_f1)):
        m = cls(task="binary")
        half = n // 3
        m.update(sd[:half], td[:half])                       # two updates: states accumulate across batches
        m.update(sd[half:], td[half:])
        assert abs(float(m.compute()) - ref(s, t)) < 1e-5, cls.__name__
        m.reset()
        m.update(sd[:half], td[:half])
        assert abs(float(m.compute()) - ref(s[:half], t[:half])) < 1e-5
    # degenerate: one class only
    ones = torch.ones(10, dtype=torch.int32, device=DEV)
    a = M.AUROC(task="binary"); a.update(sd[:10], ones)
    assert float(a.compute()) == 0.0 == O.binary_auroc(s[:10], np.ones(10))


def test_bootstrapped_collection_with_injected_poisson_counts():
    from biomedkg_amd.utils import metrics as M
    from oracle import metrics as O
    batches = [_scores(4000, 0.4, True, 1), _scores(2500, 0.6, False, 2)]
    gen = np.random.default_rng(7)
    counts = [[gen.poisson(1.0, size=len(b[0])) for b in batches] for _ in range(10)]   # [bootstrap][update]
    calls = {"u": -1, "seen": set()}

    def poisson(n, device, b):
        if b == 0:
            calls["u"] += 1
        return torch.from_numpy(counts[b][calls["u"] % len(batches)]).to(device)
    coll = M.MetricCollection({"AUROC": M.BootStrapper(M.AUROC(task="binary")),
                               "AveragePrecision": M.BootStrapper(M.AveragePrecision(task="binary")),
                               "F1": M.BootStrapper(M.F1Score(task="binary"))}).clone(prefix="val_")
    M.BootStrapper.poisson = staticmethod(poisson)
    try:
        for name, metric in coll.items():                    # one metric at a time so that the update counter is per metric
            calls["u"] = -1
            for s, t in batches:
                metric.update(torch.from_numpy(s).to(DEV), torch.from_numpy(t).to(DEV).to(torch.int32))
    finally:
        M.BootStrapper.poisson = None
    got = coll.compute()
    assert set(got) == {f"val_{k}_{s}" for k in ("AUROC", "AveragePrecision", "F1") for s in ("mean", "std")}
    for key, fn in (("AUROC", O.binary_auroc), ("AveragePrecision", O.binary_average_precision), ("F1", O.binary_f1)):
        mean, std = O.bootstrap(fn, batches, counts)
        assert abs(float(got[f"val_{key}_mean"]) - mean) < 1e-5, key
        assert abs(float(got[f"val_{key}_std"]) - std) < 1e-5, key


@pytest.mark.parametrize("name", ["rgcn_dismult_neg3", "rgcn_transe_gcl_fixid"])
def test_validation_step_end_to_end_on_a_golden_case(name):
    """validation_step on a golden KGE case: the scores reproduce the reference module's own (tests/golden/kge_steps.pt),
    and the metrics logged by validation_step / on_validation_epoch_end are the torchmetrics definitions applied to the
    scores the step produced (rank metrics over 720 near-zero random-init scores flip with the last bit, so the checker
    is fed the step's own scores, which are held to the golden ones separately)."""
    from types import SimpleNamespace
    from biomedkg_amd import rand as R
    from biomedkg_amd.kge_module import KGEModule
    from biomedkg_amd.utils import metrics as M
    from oracle import metrics as O
    g = load_golden("kge_steps")[name]
    mod = KGEModule(**g["hparams"]).to(DEV)
    mod.load_state_dict(g["state"], strict=True)
    mod.train()                                              # the golden step was recorded with its dropout draws
    if g["fix_edge_id"] is not None:
        mod.fix_edge_id = g["fix_edge_id"]
    mapping = {i: f"rel{i}" for i in range(g["hparams"]["num_relation"])}
    mod.edge_mapping = mapping
    b = SimpleNamespace(x=g["x"].to(DEV), edge_index=g["edge_index"].to(DEV), edge_type=g["edge_type"].to(DEV))
    n_pred = g["pos_pred"].numel() + g["neg_pred"].numel()
    gen = np.random.default_rng(3)
    counts = [[gen.poisson(1.0, size=n_pred)] for _ in range(10)]
    M.BootStrapper.poisson = staticmethod(lambda n, device, bi: torch.from_numpy(counts[bi][0]).to(device))
    seen = {}
    upd = mod.valid_metrics.update
    mod.valid_metrics.update = lambda p_, t_: (seen.update(pred=p_.detach().cpu(), gt=t_.detach().cpu()), upd(p_, t_))[1]
    try:
        with R.use(R.ReplayRand(g["tape"], DEV)):
            loss = mod.validation_step(b, 0)
        mod.on_validation_epoch_end()
    finally:
        M.BootStrapper.poisson = None
    torch.testing.assert_close(loss.detach().cpu(), g["loss"], rtol=0, atol=1e-5)
    logged = {k: float(v) for k, v in mod.logged_metrics.items()}
    torch.testing.assert_close(seen["pred"], torch.cat([g["pos_pred"], g["neg_pred"]]), rtol=1e-4, atol=1e-5)
    pred = seen["pred"].numpy()
    gt = seen["gt"].numpy().astype(np.int64)
    assert gt.tolist() == [1] * g["pos_pred"].numel() + [0] * g["neg_pred"].numel()
    for key, fn in (("AUROC", O.binary_auroc), ("AveragePrecision", O.binary_average_precision), ("F1", O.binary_f1)):
        mean, std = O.bootstrap(fn, [(pred, gt)], counts)
        assert abs(logged[f"val_{key}_mean"] - mean) < 1e-4, (key, logged[f"val_{key}_mean"], mean)
        assert abs(logged[f"val_{key}_std"] - std) < 1e-4, key
    et = g["edge_type"] if g["fix_edge_id"] is None else torch.full_like(g["edge_type"], g["fix_edge_id"])
    for k, v in O.edge_wise_precision(g["pos_pred"].numpy(), et.numpy(), mapping).items():
        assert abs(logged[k] - v) < 1e-6, k
