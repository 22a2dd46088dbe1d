"""-m gpu: one training / evaluation step through data.py -> engine.py -> the HIP model, losses and evaluator,
against the same step computed by the CPU oracles on the same batch."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, model_pair
from oracle import loss_oracle as LO
from echoglad_amd import data, engine, evaluators, losses

pytestmark = pytest.mark.gpu


def _setup(frame, naux, coord, seed):
    hip, ref = model_pair(frame, naux, 2, coord=coord, seed=seed)
    torch.manual_seed(seed)
    emb_ref = torch.nn.Conv2d(1, 128, kernel_size=1)
    emb_hip = torch.nn.Conv2d(1, 128, kernel_size=1).to(DEV)
    emb_hip.load_state_dict(emb_ref.state_dict())
    np.random.seed(seed)
    ds = data.SyntheticEchoDataset(num_aux_graphs=naux, frame_size=frame, use_coordinate_graph=coord)
    return hip, ref, emb_hip, emb_ref, ds


@pytest.mark.parametrize("frame,naux,coord", [(16, 3, False), (32, 4, True)])
def test_train_step_losses_and_gradients_match_cpu_step(frame, naux, coord):
    B = 2
    hip, ref, emb_hip, emb_ref, ds = _setup(frame, naux, coord, 3)
    batch = data.collate([ds[i] for i in range(B)], ds.topology)
    # p = 0 dropout and train-mode BN: the only stochastic piece of the reference's train mode is switched off
    for m in list(hip.modules()) + list(ref.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train(); ref.train()
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1),
            "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux)}
    if coord:
        crit["coordinate"] = engine.MSE(1)
    params = list(hip.parameters()) + list(emb_hip.parameters())
    opt = torch.optim.SGD(params, lr=0.0)                                      # lr 0: gradients stay inspectable
    import copy
    dev_batch = data.to_device(copy.copy(batch), DEV)
    loss, parts, preds, _ = engine.train_step({"embedder": emb_hip, "landmark": hip}, dev_batch, crit, opt, B, coord)
    # the same step on the CPU: oracle model + oracle losses
    x = emb_ref(batch.x.cpu())
    nc = batch.node_coords.cpu().clone() if coord else None
    out = ref(x=x, node_coords=nc, edge_index=batch.edge_index.cpu(), batch_idx=batch.batch.cpu(), node_type=batch.node_type.cpu())
    p_ref, c_ref = out
    n = p_ref.shape[0] // B
    want = LO.weighted_bce_with_logits(p_ref.view(B, n, 4), batch.y.cpu().view(B, n, 4), batch.valid_labels.cpu(), 9000, 1) + \
        LO.expected_landmark_mse(p_ref, batch.y.cpu(), batch.valid_labels.cpu(), B, frame, naux, loss_weight=10)
    if coord:
        want = want + torch.nn.functional.mse_loss(c_ref, batch.node_coord_y.cpu())
    want.backward()
    assert abs(float(loss) - float(want.detach())) <= 2e-4 * abs(float(want.detach()))
    assert float((preds.cpu() - p_ref.detach()).abs().max()) < 2e-4
    ref_grads = dict(ref.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for name, p in hip.named_parameters():
        g_ref = ref_grads[name].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        # GCN biases in front of a train-mode BatchNorm have a mathematically zero gradient: absolute floor
        err = float((p.grad.cpu() - g_ref).abs().max())
        assert err <= 2e-2 * float(g_ref.abs().max()) + 1e-4 * gmax, (name, err, float(g_ref.abs().max()), gmax)
    ge = emb_hip.weight.grad.cpu()
    assert float((ge - emb_ref.weight.grad).abs().max()) <= 2e-2 * float(emb_ref.weight.grad.abs().max()) + 1e-6


def test_training_reduces_the_loss_and_eval_step_feeds_the_evaluator():
    B, frame, naux = 2, 16, 3
    hip, _, emb_hip, _, ds = _setup(frame, naux, False, 5)
    batch = data.to_device(data.collate([ds[i] for i in range(B)], ds.topology), DEV)
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux)}
    model = {"embedder": emb_hip, "landmark": hip}
    opt = torch.optim.Adam(list(hip.parameters()) + list(emb_hip.parameters()), lr=1e-3)
    hip.train()
    first = last = None
    for it in range(12):
        loss, _, _, _ = engine.train_step(model, batch, crit, opt, B)
        first = float(loss) if first is None else first
        last = float(loss)
    assert last < 0.95 * first, (first, last)
    hip.eval()
    ev = {"landmark": evaluators.LandmarkExpectedCoordiantesEvaluator(None, B, frame, False)}
    preds, _, ls = engine.eval_step(model, batch, crit, B, evaluators=ev)
    rec = ev["landmark"].get_last()
    want = LO.evaluate_landmarks(preds.cpu(), batch.y.cpu(), batch.pix2mm_x.cpu(), batch.pix2mm_y.cpu(), batch.valid_labels.cpu(), B, frame)
    for k, v in rec.items():
        # degenerate synthetic labels (two landmarks on one pixel) give inf / nan percentages in the reference too
        assert np.isclose(float(v), float(want[k]), rtol=1e-4, atol=1e-4, equal_nan=True), (k, float(v), float(want[k]))
    assert set(ls) == {"bce", "elm"}


@pytest.mark.parametrize("frame,naux,coord,p", [(32, 4, True, 0.5), (16, 3, False, 0.5), (32, 4, True, 0.0), (224, 7, True, 0.5)])
def test_graphed_train_step_replays_equal_eager_steps_under_the_same_epoch(frame, naux, coord, p):
    """engine.GraphedTrainStep: the whole step (embedder, model, three criteria, backward, Adam) as ONE HIP graph.  Replay k must be
    the eager step that draws the same host seeds and runs under dropout epoch k -- bit for bit: same kernels, same arguments but
    the epoch word the graph's first node bumps.  With p = 0.5 consecutive replays must also differ from each other's masks
    (the epoch does reach the kernels), and the epoch must have advanced by one per replay."""
    import copy
    from echoglad_amd import ops
    B, warm, replays = 2, 2, 3

    def build():
        hip, _, emb_hip, _, ds = _setup(frame, naux, coord, 11)
        for m in hip.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = p
        hip.train()
        batch = data.to_device(data.collate([ds[i] for i in range(B)], ds.topology), DEV)
        crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux)}
        if coord:
            crit["coordinate"] = engine.MSE(1)
        # (the embedder -- a torch Conv2d, outside the hot path -- stays frozen: MIOpen's weight-gradient kernel adds with atomics,
        #  its last bits differ from run to run, eagerly as well: tools/dbg_graph_step.py)
        for q in emb_hip.parameters():
            q.requires_grad_(False)
        params = list(hip.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
        model = {"embedder": emb_hip, "landmark": hip}
        coords0 = batch.node_coords.clone() if coord else None

        def loss_fn():
            if coord:
                batch.node_coords = coords0.clone()                      # (the model updates the landmark guesses it is given)
            preds, coord_preds = engine.forward_batch(model, batch, coord)
            ls = engine.compute_loss(crit, preds, batch.y, coord_preds, batch.node_coord_y if coord else None, batch.valid_labels, B)
            return sum(ls.values()), preds
        return hip, emb_hip, opt, loss_fn, params

    ops.dropout_epoch_set(0)
    # --- the graph: `warm` eager steps, then the capture (which draws the seeds every replay reuses), then `replays` replays
    torch.manual_seed(123)
    hip_g, emb_g, opt_g, loss_g, params_g = build()
    torch.manual_seed(77)
    step = engine.GraphedTrainStep(loss_g, opt_g, warmup=warm)
    e0 = ops.dropout_epoch()                                              # (the capture itself executes nothing)
    losses_g, preds_g = [], []
    for _ in range(replays):
        out = step()
        losses_g.append(float(out[0]))
        preds_g.append(out[1].clone())
    assert ops.dropout_epoch() == e0 + replays and step.replays == replays
    # --- the same thing eagerly: `warm` steps, then `replays` steps that all start from the host RNG state the capture started from
    ops.dropout_epoch_set(e0)
    torch.manual_seed(123)
    hip_e, emb_e, opt_e, loss_e, params_e = build()
    torch.manual_seed(77)

    def eager():
        out = loss_e()
        opt_e.zero_grad(set_to_none=True)
        out[0].backward()
        opt_e.step()
        return out
    for _ in range(warm):
        eager()
    rng = torch.get_rng_state()
    for k in range(replays):
        torch.set_rng_state(rng)
        ops.dropout_epoch_set(e0 + k + 1)
        out = eager()
        assert float(out[0].detach()) == losses_g[k], (k, float(out[0].detach()), losses_g[k])
        assert torch.equal(out[1].detach(), preds_g[k]), k
    for a, b in zip(params_g, params_e):
        assert torch.equal(a.detach(), b.detach())
    for (na, ba), (nb, bb) in zip(hip_g.named_buffers(), hip_e.named_buffers()):
        assert na == nb and torch.equal(ba, bb), na
    if p > 0:
        assert len(set(losses_g)) == replays, losses_g                  # fresh masks at every replay
    ops.dropout_epoch_set(0)
    with pytest.raises(ValueError):
        engine.GraphedTrainStep(loss_e, torch.optim.Adam(params_e, lr=1e-3), warmup=1)      # (not capturable)


def test_to_device_moves_the_constant_tensors_of_a_topology_once():
    """data.collate + data.to_device: edge_index / batch / node_type of a (topology, batch size) are the SAME device tensors for every
    batch (one host-to-device copy per device; the model resolves the edge_index by identity), the per-sample tensors are fresh."""
    _, _, _, _, ds = _setup(16, 3, False, 9)
    a = data.to_device(data.collate([ds[0], ds[1]], ds.topology), DEV)
    b = data.to_device(data.collate([ds[2], ds[3]], ds.topology), DEV)
    assert a.edge_index is b.edge_index and a.batch is b.batch and a.node_type is b.node_type and a.edge_index.is_cuda
    assert a.x is not b.x and a.x.is_cuda and not torch.equal(a.x, b.x)
    n = ds.topology.num_nodes
    assert torch.equal(a.edge_index.cpu(), torch.cat([ds.edge_index, ds.edge_index + n], dim=1))


def test_documented_graphed_loop_takes_new_batches_without_touching_the_graph_tensors():
    """``copy_batch_(static, batch); step()``.  New frames / labels arrive in place and reach the replay (the
    loss follows the batch), while edge_index / batch / node_type are neither copied (no version bump, no pageable host-to-device
    copy per step) nor silently replaced: another graph is refused."""
    import copy
    B, frame, naux = 2, 32, 4
    hip, _, emb_hip, _, ds = _setup(frame, naux, True, 13)
    for m in hip.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train()
    for q in emb_hip.parameters():
        q.requires_grad_(False)
    host = [data.collate([ds[2 * i], ds[2 * i + 1]], ds.topology) for i in range(3)]
    static = data.to_device(copy.copy(host[0]), DEV)
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux),
            "coordinate": engine.MSE(1)}
    opt = torch.optim.Adam(list(hip.parameters()), lr=0.0, capturable=True)         # lr 0: the loss is a function of the batch alone
    model = {"embedder": emb_hip, "landmark": hip}
    coords0 = static.node_coords.clone()

    def loss_fn():
        static.node_coords = coords0.clone()
        preds, cp = engine.forward_batch(model, static, True)
        return sum(engine.compute_loss(crit, preds, static.y, cp, static.node_coord_y, static.valid_labels, B).values())

    step = engine.GraphedTrainStep(loss_fn, opt, warmup=1)
    ei, v0 = static.edge_index, static.edge_index._version
    seen = []
    for k in (1, 2, 1):
        data.copy_batch_(static, host[k])
        seen.append(float(step()[0]))
        assert torch.equal(static.x.cpu(), host[k].x) and torch.equal(static.y.cpu(), host[k].y)
    assert static.edge_index is ei and ei._version == v0
    assert seen[0] != seen[1] and seen[0] == seen[2]             # the replay reads the new batch; same batch, same loss (p = 0, lr = 0)
    other = data.collate([ds[0], ds[1]])                         # fresh tensors with another graph in them
    other.edge_index = other.edge_index.flip(0).contiguous()
    fresh_static = data.to_device(copy.copy(data.collate([ds[0], ds[1]])), DEV)
    with pytest.raises(ValueError):
        data.copy_batch_(fresh_static, other)


@pytest.mark.gpu
@pytest.mark.parametrize("wd,maximize", [(0.0, False), (0.01, False), (0.0, True)])
def test_one_launch_adam_equals_torch_adam(wd, maximize):
    """echoglad_amd.optim.Adam (eg_adam_step: the whole parameter list in one launch, the step count a device float advanced by the kernel)
    against torch.optim.Adam on the same gradients, 5 steps: parameters and both moments to a few ulps (the same formulas; torch's
    foreach form orders two multiplications differently); state_dict round trip; more tensors than one call takes."""
    from echoglad_amd.optim import Adam
    rs = np.random.RandomState(3)
    shapes = [(128, 128), (128,), (32, 136), (2, 16), (1,), (4097,)] + [(7,)] * 100          # 106 tensors: two calls per step
    ours = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32)).to(DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a = Adam(ours, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, maximize=maximize)
    b = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, maximize=maximize)
    for it in range(5):
        for p, q in zip(ours, ref):
            g = torch.from_numpy(rs.standard_normal(tuple(p.shape)).astype(np.float32)).to(DEV) * (0.1 + it)
            p.grad, q.grad = g.clone(), g.clone()
        if it == 2:
            ours[3].grad = None                       # a parameter without a gradient in one step: skipped by both
            ref[3].grad = None
        a.step()
        b.step()
        for k, (p, q) in enumerate(zip(ours, ref)):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), (it, k)
    assert float(a.state[ours[0]]["step"]) == 5.0
    for p, q in zip(ours, ref):
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.allclose(a.state[p][key], b.state[q][key], rtol=1e-5, atol=1e-6), key
    # state_dict round trip into a fresh optimizer (and torch's own state loads as well): the next step is the same step
    import copy
    sd = copy.deepcopy(a.state_dict())                 # (load_state_dict keeps tensors that need no cast: without the copy the two would share moments)
    ours2 = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a2 = Adam(ours2, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, maximize=maximize)
    a2.load_state_dict(sd)
    ours3 = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a3 = Adam(ours3, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, maximize=maximize)
    a3.load_state_dict(copy.deepcopy(b.state_dict()))
    for p, p2, p3, q in zip(ours, ours2, ours3, ref):
        g = torch.from_numpy(rs.standard_normal(tuple(p.shape)).astype(np.float32)).to(DEV)
        p.grad, p2.grad, p3.grad, q.grad = g.clone(), g.clone(), g.clone(), g.clone()
    a.step(); a2.step(); a3.step(); b.step()
    for k, (p, p2, p3, q) in enumerate(zip(ours, ours2, ours3, ref)):
        assert torch.equal(p, p2), k
        assert float((p3 - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), k
    assert float(a2.state[ours2[0]]["step"]) == 6.0 and float(a3.state[ours3[0]]["step"]) == 6.0
    cpu = torch.nn.Parameter(torch.zeros(3))
    cpu.grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        Adam([cpu]).step()


@pytest.mark.gpu
def test_graphed_train_step_with_the_one_launch_adam():
    """engine.GraphedTrainStep with echoglad_amd.optim.Adam: the update is part of the captured graph and its step count advances with
    every replay (bias corrections change from replay to replay)."""
    from echoglad_amd.optim import Adam
    w = torch.nn.Parameter(torch.ones(64, 64, device=DEV))
    x = torch.randn(8, 64, device=DEV)
    opt = Adam([w], lr=1e-2)
    ref_w = torch.nn.Parameter(w.detach().clone())
    ref_opt = torch.optim.Adam([ref_w], lr=1e-2)
    step = engine.GraphedTrainStep(lambda: ((x @ w) ** 2).mean(), opt, warmup=2)
    for _ in range(4):
        step()
    for _ in range(6):
        ref_opt.zero_grad()
        ((x @ ref_w) ** 2).mean().backward()
        ref_opt.step()
    assert float(opt.state[w]["step"]) == 6.0
    assert float((w - ref_w).abs().max()) <= 1e-5
    # a scheduler under the captured step: lr as a device scalar, changed between replays (the reference's default schedule is
    # ReduceLROnPlateau, which fill_()s a tensor lr)
    w2 = torch.nn.Parameter(torch.ones(64, 64, device=DEV))
    ref2 = torch.nn.Parameter(w2.detach().clone())
    lr = torch.tensor(1e-2, device=DEV)
    opt2 = Adam([w2], lr=lr, weight_decay=1e-4)
    ref_opt2 = torch.optim.Adam([ref2], lr=1e-2, weight_decay=1e-4)
    step2 = engine.GraphedTrainStep(lambda: ((x @ w2) ** 2).mean(), opt2, warmup=1)
    for k in range(5):
        if k == 2:
            lr.fill_(1e-3)
            ref_opt2.param_groups[0]["lr"] = 1e-3
        step2()
    for k in range(6):
        ref_opt2.param_groups[0]["lr"] = 1e-2 if k < 3 else 1e-3      # (1 warm-up step + replays 0, 1 at 1e-2; replays 2 - 4 at 1e-3)
        ref_opt2.zero_grad()
        ((x @ ref2) ** 2).mean().backward()
        ref_opt2.step()
    assert float((w2 - ref2).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_eval_after_graphed_training_sees_the_trained_state():
    """A replayed training step runs no host code -- no in-place version counter moves -- and the one-launch Adam writes the parameters from
    a kernel of its own: the model's caches of folded inference parameters must not survive either.  Inference after graphed training
    (and after eager steps with echoglad_amd.optim.Adam) == inference of a fresh model loaded with the trained state."""
    from echoglad_amd.optim import Adam
    frame, naux, B = 16, 3, 2
    hip, _, emb, _, ds = _setup(frame, naux, True, 5)
    batch = data.to_device(data.collate([ds[i] for i in range(B)], ds.topology), DEV)
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux), "coordinate": engine.MSE(1)}
    for q in emb.parameters():
        q.requires_grad_(False)
    model = {"embedder": emb, "landmark": hip}

    def infer(m):
        m.eval()
        with torch.no_grad():
            return [t.clone() for t in engine.forward_batch({"embedder": emb, "landmark": m}, batch, True)]

    first = infer(hip)                                   # (fills the caches of folded parameters / the inference graph)
    hip.train()
    opt = Adam(list(hip.parameters()), lr=1e-2)

    def loss_fn():
        preds, cp = engine.forward_batch(model, batch, True)
        return engine.total_loss(engine.compute_loss(crit, preds, batch.y, cp, batch.node_coord_y, batch.valid_labels, B))
    step = engine.GraphedTrainStep(loss_fn, opt, warmup=2)
    for _ in range(3):
        step()
    got = infer(hip)
    assert not torch.equal(got[0], first[0])             # five steps at lr 1e-2 moved it
    fresh, _, _, _, _ = _setup(frame, naux, True, 6)
    fresh.load_state_dict({k: v.clone() for k, v in hip.state_dict().items()})
    want = infer(fresh)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # eager steps with the one-launch Adam, model left in eval mode in between (fine-tuning with frozen statistics)
    hip.eval()
    for q in hip.parameters():
        q.requires_grad_(True)
    before = infer(hip)
    opt2 = Adam(list(hip.parameters()), lr=1e-2)
    preds, cp = engine.forward_batch(model, batch, True)
    engine.total_loss(engine.compute_loss(crit, preds, batch.y, cp, batch.node_coord_y, batch.valid_labels, B)).backward()
    opt2.step()
    after = infer(hip)
    fresh.load_state_dict({k: v.clone() for k, v in hip.state_dict().items()})
    want = infer(fresh)
    assert not torch.equal(after[0], before[0]) and torch.equal(after[0], want[0])
