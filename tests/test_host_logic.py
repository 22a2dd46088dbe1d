"""CPU: host-side logic added in round 6 that needs no GPU -- the stacked layout the heads' parameters live in, the optimizer's
argument checks."""
import pytest
import torch

from echoglad_amd import nn as egnn


def test_head_parameter_bank_layout_tiles_the_flat_buffer():
    offs = egnn._head_param_offsets()
    sizes = [egnn._HEAD_SIZES[j] for _ in range(4) for j in range(10)]            # params[10 * k + j] has size _HEAD_SIZES[j]
    spans = sorted((o, o + n) for o, n in zip(offs, sizes))
    assert spans[0][0] == 0 and spans[-1][1] == 4 * sum(egnn._HEAD_SIZES)
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))                   # no gap, no overlap
    # the 4 heads' slices of one array are consecutive: the stacked array the kernels take is ONE slice of the bank
    for j in range(10):
        assert [offs[10 * k + j] for k in range(4)] == [offs[j] + k * egnn._HEAD_SIZES[j] for k in range(4)]


def test_move_into_and_views_of():
    ts = [torch.arange(6.0).view(2, 3), torch.arange(4.0) + 10, torch.tensor([7.0])]
    offs, bank, moved = [0, 6, 10], torch.empty(11), {}
    assert not egnn._views_of(bank, ts, offs)
    egnn._move_into(bank, ts, offs, lambda i, v: moved.__setitem__(i, v))
    views = [moved[i] for i in range(3)]
    assert egnn._views_of(bank, views, offs)
    for t, v in zip(ts, views):
        assert v.shape == t.shape and torch.equal(v, t)
    assert torch.equal(bank, torch.cat([t.reshape(-1) for t in ts]))
    views[1] = views[1].clone()                                                  # somebody re-allocated one of them
    assert not egnn._views_of(bank, views, offs)
    assert not egnn._views_of(bank, [v.double() for v in moved.values()], offs)  # wrong dtype


def test_adam_argument_checks_and_no_cpu_path():
    from echoglad_amd.optim import Adam
    p = torch.nn.Parameter(torch.zeros(3))
    for kw in (dict(lr=-1.0), dict(eps=-1e-8), dict(betas=(1.0, 0.999)), dict(betas=(0.9, -0.1)), dict(weight_decay=-1.0)):
        with pytest.raises(ValueError):
            Adam([p], **kw)
    opt = Adam([p], lr=1e-3)
    assert opt.param_groups[0]["capturable"] and opt.param_groups[0]["fused"]    # (what engine.GraphedTrainStep looks for)
    opt.step()                                                                   # no gradient anywhere: nothing to do, nothing launched
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        opt.step()                                                               # a CPU parameter: refused before any launch
