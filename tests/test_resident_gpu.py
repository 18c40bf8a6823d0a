""" The LDS-resident forward (`dcv_resident_forward`, deepcv_amd/resident.py) against the per-layer kernels and the CPU oracle: same model, same seed, same
batch.  The per-layer path is itself oracle-checked (tests/test_model_gpu.py), so agreement to rounding here pins the resident kernel; the oracle comparison
at the bench batch size is repeated directly. """
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED, DATA_SEED = 563454, 434546


def _model(params, spec, device):
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    torch.manual_seed(SEED)
    return create_model({'trainset': SyntheticImageDataset(8)}, params[spec]).to(device)


def _assert_replay_mechanism(step, list_calls: int):
    """ The captured step replays as a verified launch list of `list_calls` library calls — or, with `DCV_REPLAY_LIST=0` (the documented switch for the hipGraph
    replay), as its hipGraph: either way the mechanism the environment asked for, so the suite is green in both modes. """
    if step._replay_list:
        assert step._launch_list is not None and len(step._launch_list.calls) == list_calls, \
            [getattr(fn, '__name__', repr(fn)) for fn, _, _ in getattr(step._launch_list, 'calls', [])]
    else:
        assert step._launch_list is None and step.graph_fb is not None


def _run(model, x, y, resident_on, backward_on=False):
    from deepcv_amd import modules, resident
    prev, prev_b = resident.set_enabled(resident_on), resident.set_backward_enabled(backward_on)
    try:
        before, before_b = resident.launches(), resident.backward_launches()
        model.train()
        logits = model(x)
        loss = modules.CrossEntropyLoss()(logits, y)
        loss.backward()
        torch.cuda.synchronize()
        resident.check_status()
        assert resident.backward_launches() - before_b == (1 if (resident_on and backward_on) else 0), 'the persistent backward launch did not run'
        return logits.detach().clone(), loss.item(), resident.launches() - before
    finally:
        resident.set_enabled(prev)
        resident.set_backward_enabled(prev_b)


@pytest.mark.parametrize('backward_on', [False, True], ids=['forward', 'forward+backward'])
@pytest.mark.parametrize('bs', [2, 7, 64, 511, 512])
def test_resident_forward_matches_per_layer_kernels(device, params, bs, backward_on):
    """ Logits, loss, every parameter gradient (the backward pass reads the tensors the resident launch saved: y, layer outputs, BatchNorm / GroupNorm
    statistics; with `backward_on` its serial chain is the second persistent launch) and the running statistics, against the per-layer path.  Odd batch
    sizes leave a block with one sample. """
    from deepcv_amd import resident
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(bs, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (bs,), generator=g).to(device)
    a, b = _model(params, 'image_classifier_model', device), _model(params, 'image_classifier_model', device)
    prev_cap, resident.capture_programs = resident.capture_programs, True
    try:
        la, loss_a, n_a = _run(a, x, y, True, backward_on)
    finally:
        resident.capture_programs = prev_cap
    if backward_on:
        # the head's weight gradient is part of the launch from 320 samples on (one block per 8 of its 1280 input positions), its own launch below
        head = resident.last_programs['backward'][0].head
        assert head.enabled == 1 and bool(head.fc_dw) == (bs >= 320)
    lb, loss_b, n_b = _run(b, x, y, False)
    assert n_a == 1, f'the resident path did not run: {resident.why_not(a, x)}'
    assert n_b == 0
    scale = lb.abs().max().item()
    assert (la - lb).abs().max().item() <= 2e-6 * max(scale, 1.0), (la - lb).abs().max().item()
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_b)
    # large batches: the two forwards differ in the last bit of the BatchNorm sums, 2e6 pre-activations per layer put about one ReLU within that distance of
    # its kink, and one flipped pixel moves a weight gradient (a sum of 5e5 terms of random sign) by ~1e-3 of its size (tests/test_model_gpu.py)
    rel = 2e-4 if bs <= 64 else 1e-2
    gscale = max(p.grad.abs().max().item() for p in b.parameters())
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        err = (pa.grad - pb.grad).abs().max().item()
        assert err <= rel * pb.grad.abs().max().item() + 2e-6 * gscale, f'{name}: {err:.3e} (max {pb.grad.abs().max().item():.3e})'
    for (name, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-5, atol=1e-7), name


@pytest.mark.parametrize('bs', [7, 64])
def test_general_paths_on_a_3x3_variant_of_the_stock_model(device, params, bs):
    """ The stock shapes run constant-extent paths in both persistent launches (5x5 layers on 32 x 32 maps: matrix-core weight gradient, fused moments, fixed
    tile fills and apply steps); every other shape runs the general code those paths bypass.  The same stack with 3x3 filters in its first three layers
    is such a shape (3x3 convolution / data-gradient variants, the VALU weight gradient, general apply / dz / moments passes): resident launches against the
    per-layer kernels, as for the stock model. """
    import copy
    from deepcv_amd import resident
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    hp = copy.deepcopy(params['image_classifier_model'])
    convs = [m['conv2d'] for m in hp['architecture'][0]['_nested_deepcvmodule']['architecture'] if isinstance(m, dict) and 'conv2d' in m]
    assert len(convs) == 5
    for c in convs[:3]:
        assert tuple(c['kernel_size']) == (5, 5)
        c['kernel_size'], c['padding'] = (3, 3), 1

    def build():
        torch.manual_seed(SEED)
        return create_model({'trainset': SyntheticImageDataset(8)}, hp).to(device)
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(bs, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (bs,), generator=g).to(device)
    a, b = build(), build()
    la, loss_a, n_a = _run(a, x, y, True, True)
    lb, loss_b, n_b = _run(b, x, y, False)
    assert n_a == 1, f'the resident path did not run: {resident.why_not(a, x)}'
    assert n_b == 0
    assert (la - lb).abs().max().item() <= 2e-6 * max(lb.abs().max().item(), 1.0)
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_b)
    gscale = max(p.grad.abs().max().item() for p in b.parameters())
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        err = (pa.grad - pb.grad).abs().max().item()
        assert err <= 2e-4 * pb.grad.abs().max().item() + 2e-6 * gscale, f'{name}: {err:.3e} (max {pb.grad.abs().max().item():.3e})'
    for (name, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-5, atol=1e-7), name


@pytest.mark.parametrize('offset', [0.0, 12.0])
def test_channels_far_from_zero_against_the_oracle_in_double_precision(device, params, offset):
    """ The backward launch derives a sample's sums of u, u^2 (GroupNorm algebra) from the forward's fp32 sums of y, y^2; with a channel mean many standard
    deviations from zero (every convolution bias moved by `offset`: 20 - 40 sigma of the pre-activations) that subtraction cancels.  The
    referee is the oracle in DOUBLE precision: the derived sums stay consistent with the forward's GroupNorm statistics (same inputs), and the launches
    end up closer to it than the per-layer kernels (tools/offset_err.py: 1.5e-4 against 7.3e-4 of a tensor's largest gradient at offset 12). """
    import copy
    from test_model_gpu import _build
    from deepcv_amd import resident
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(64, 3, 32, 32, generator=g), torch.randint(0, 10, (64,), generator=g)
    a, oracle = _build(params, 'image_classifier_model', 10)
    b = copy.deepcopy(a)
    a, b, oracle = a.to(device), b.to(device), oracle.double()
    shifted = 0
    for m in (a, b, oracle):
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d) and mod.bias is not None:
                with torch.no_grad(): mod.bias.add_(offset)
                shifted += 1
    assert shifted == 3 * 5
    oracle.train()
    torch.nn.CrossEntropyLoss()(oracle(x.double()), y).backward()
    _, _, n_a = _run(a, x.to(device), y.to(device), True, True)
    _run(b, x.to(device), y.to(device), False)
    assert n_a == 1, f'the resident path did not run: {resident.why_not(a, x.to(device))}'
    gscale = max(p.grad.abs().max().item() for p in oracle.parameters())
    for tag, m, rel in (('persistent launches', a, 2e-4 if offset == 0.0 else 1e-3), ('per-layer kernels', b, 2e-4 if offset == 0.0 else 5e-3)):
        for (name, pm), (_, po) in zip(m.named_parameters(), oracle.named_parameters()):
            err = (pm.grad.double().cpu() - po.grad).abs().max().item()
            assert err <= rel * max(po.grad.abs().max().item(), 1e-2 * gscale), f'{tag}, {name}: {err:.3e} (max {po.grad.abs().max().item():.3e}, model {gscale:.3e})'


def test_weight_gradients_outside_the_backward_launch(device, params):
    """ `DCV_RESIDENT_WGRAD=0`: the persistent backward launch leaves the weight gradients to the multi-layer launch, which reads its dz tensors. """
    from deepcv_amd import resident
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)
    a, b = _model(params, 'image_classifier_model', device), _model(params, 'image_classifier_model', device)
    prev, resident._wgrad_in_launch = resident._wgrad_in_launch, False
    try:
        la, loss_a, _ = _run(a, x, y, True, True)
    finally:
        resident._wgrad_in_launch = prev
    lb, loss_b, _ = _run(b, x, y, True, True)
    assert loss_a == loss_b
    gscale = max(p.grad.abs().max().item() for p in b.parameters())
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa.grad - pb.grad).abs().max().item() <= 2e-4 * pb.grad.abs().max().item() + 2e-6 * gscale, name


def test_resident_forward_matches_oracle_at_the_bench_batch(device, params):
    """ BASELINE.json bars on the resident path itself: logits rtol 1e-3, loss rtol 1e-4 against the CPU oracle at bs 512. """
    from deepcv_amd import resident
    from oracle import deepcv_oracle as O
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(512, 3, 32, 32, generator=g), torch.randint(0, 10, (512,), generator=g)
    model = _model(params, 'image_classifier_model', device)
    torch.manual_seed(SEED)
    oracle = O.create_model((3, 32, 32), params['image_classifier_model'], 10)
    oracle.train()
    lo = oracle(x)
    loss_o = torch.nn.functional.cross_entropy(lo, y).item()
    ln, loss_n, n = _run(model, x.to(device), y.to(device), True, True)
    assert n == 1, resident.why_not(model, x.to(device))
    rel = ((ln.cpu() - lo.detach()).abs().max() / lo.detach().abs().max()).item()
    assert rel <= 1e-3 and abs(loss_n - loss_o) <= 1e-4 * abs(loss_o), (rel, loss_n, loss_o)


def test_models_the_kernel_does_not_cover_fall_back(device, params):
    """ A wide spec is not a resident program: the recording aborts before anything is launched, the verdict is cached, the per-layer kernels serve it. """
    from deepcv_amd import modules, resident
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    torch.manual_seed(SEED)
    model = create_model({'trainset': SyntheticImageDataset(4, num_classes=100)}, params['cifar100_deep_model']).to(device)
    x, y = torch.randn(4, 3, 32, 32, device=device), torch.randint(0, 100, (4,), device=device)
    before = resident.launches()
    model.train()
    rm0 = [b.clone() for b in model.buffers()]
    loss = modules.CrossEntropyLoss()(model(x), y)
    loss.backward()
    torch.cuda.synchronize()
    assert resident.launches() == before and resident.why_not(model, x) is not None
    assert torch.isfinite(loss).item() and all(p.grad is not None for p in model.parameters())
    # the aborted recording touched no running statistic: exactly ONE update happened
    nbt = [b for n, b in model.named_buffers() if n.endswith('num_batches_tracked')]
    assert all(int(b) == 1 for b in nbt)
    assert not resident.wanted(model, x)          # cached verdict: no second recording attempt
    del rm0


def test_replayed_graph_on_the_resident_path_tracks_eager(device, params):
    """ The persistent launch inside a captured step: 4 replays against 4 eager steps, bit for bit (the launch generation lives in device memory). """
    from deepcv_amd import modules, optim, resident
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    batches = [(torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)) for _ in range(6)]
    curves = []
    for graph in (True, False):
        model = _model(params, 'image_classifier_model', device)
        opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
        losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
        step = GraphedTrainStep(model, opt, losses, weights)
        before = resident.launches()
        got = []
        for xb, yb in batches:
            out, _ = step(xb, yb) if graph else step._eager(xb, yb)
            got.append(out['main_loss'].item())
        torch.cuda.synchronize()
        resident.check_status()
        assert resident.launches() > before
        if graph:
            # the step as a verified list of launches: forward, cross-entropy, the head's weight gradient (its own launch below 320 samples), backward,
            # slab sums, AdamW + filters
            _assert_replay_mechanism(step, 6)
        curves.append(got)
    assert curves[0] == curves[1], curves


def test_backward_launch_flushed_in_the_middle_of_the_pass(device, params):
    """ Small gradient buckets of the data-parallel reducer fill up while the backward pass is still recording layers: the bucket pack flushes the
    pending work, i.e. launches the persistent backward launch with the layers recorded so far.  Their last data gradient then has to reach HBM, the
    remaining layers run on the per-layer kernels: same gradients as without a reducer. """
    from deepcv_amd import modules, resident
    from deepcv_amd.parallel import GradientAllReducer
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)
    a, b = _model(params, 'image_classifier_model', device), _model(params, 'image_classifier_model', device)
    reducer = GradientAllReducer(a.parameters(), bucket_cap_mb=0.002, overlap=True)         # ~500 floats per bucket: most layers get a bucket of their own
    assert len(reducer.buckets) > 4
    try:
        la, loss_a, _ = _run_plain(a, x, y)
        reducer.finish()
    finally:
        reducer.detach()
    lb, loss_b, _ = _run_plain(b, x, y)
    assert loss_a == loss_b
    gscale = max(p.grad.abs().max().item() for p in b.parameters())
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        err = (pa.grad - pb.grad).abs().max().item()
        assert err <= 2e-4 * pb.grad.abs().max().item() + 2e-6 * gscale, f'{name}: {err:.3e}'


def test_gradient_accumulation_over_two_backward_passes(device, params):
    """ A second backward pass onto existing `.grad`s: the engine ADDS what the backward functions return, so nothing may be handed over unfilled — the
    second pass must leave the recorded path (the plan decides that before its first layer records itself).  Sum of both passes' gradients against the
    per-layer kernels. """
    from deepcv_amd import modules, resident
    g = torch.Generator().manual_seed(DATA_SEED)
    xs = [torch.randn(64, 3, 32, 32, generator=g).to(device) for _ in range(2)]
    ys = [torch.randint(0, 10, (64,), generator=g).to(device) for _ in range(2)]
    a, b = _model(params, 'image_classifier_model', device), _model(params, 'image_classifier_model', device)
    for model, on in ((a, True), (b, False)):
        prev, prev_b = resident.set_enabled(on), resident.set_backward_enabled(on)
        try:
            model.train()
            for x, y in zip(xs, ys):
                modules.CrossEntropyLoss()(model(x), y).backward()        # no zero_grad in between
            torch.cuda.synchronize()
            resident.check_status()
        finally:
            resident.set_enabled(prev)
            resident.set_backward_enabled(prev_b)
    gscale = max(p.grad.abs().max().item() for p in b.parameters())
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        err = (pa.grad - pb.grad).abs().max().item()
        assert err <= 2e-4 * pb.grad.abs().max().item() + 2e-6 * gscale, f'{name}: {err:.3e}'


def _run_plain(model, x, y):
    from deepcv_amd import modules, resident
    prev, prev_b = resident.set_enabled(True), resident.set_backward_enabled(True)
    try:
        model.train()
        logits = model(x)
        loss = modules.CrossEntropyLoss()(logits, y)
        loss.backward()
        torch.cuda.synchronize()
        resident.check_status()
        return logits.detach().clone(), loss.item(), 0
    finally:
        resident.set_enabled(prev)
        resident.set_backward_enabled(prev_b)


def test_cross_entropy_inside_the_backward_launch(device, params):
    """ At the bench batch size a captured step is THREE launches (round 4: forward, backward, and ONE trailing launch — the ordered slab sums, AdamW and the
    filter re-layout, `dcv_adamw_step_prep_slabs` —; four until then): the loss (value, early read-back, arg-max hits, d loss / d logits) is computed by the
    head of the persistent backward launch from the forward launch's logits — the same arithmetic as the loss kernel the eager steps run, and the trailing launch
    the same arithmetic as the eager steps' three: losses and hit counts equal bit for bit.  `DCV_CE_IN_BACKWARD` off: four launches, the same numbers. """
    from deepcv_amd import modules, optim, resident
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    batches = [(torch.randn(512, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (512,), generator=g).to(device)) for _ in range(4)]
    curves = []
    for mode in ('in backward', 'own launch', 'eager'):
        prev, resident._ce_in_backward = resident._ce_in_backward, mode == 'in backward'
        try:
            model = _model(params, 'image_classifier_model', device)
            opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
            loss_fn = modules.CrossEntropyLoss()
            losses, weights = _setup_losses(loss_fn, None, device)
            step = GraphedTrainStep(model, opt, losses, weights)
            got = []
            for xb, yb in batches:
                if mode == 'eager':
                    out, _ = step._eager(xb, yb)
                    got.append((out['main_loss'].item(), int(loss_fn.last_correct.item())))
                else:
                    step(xb, yb)
                    got.append((step.loss_values()['main_loss'], int(loss_fn.last_correct.item())))
            torch.cuda.synchronize()
            resident.check_status()
            if mode != 'eager':
                _assert_replay_mechanism(step, 3 if mode == 'in backward' else 4)
            curves.append(got)
        finally:
            resident._ce_in_backward = prev
    assert curves[0] == curves[1] == curves[2], curves


def test_launch_list_replay_is_dropped_when_the_step_runs_foreign_kernels(device, params):
    """ A captured step is replayed as a list of this library's launches only after the first replay came out bit-identical through the graph and through
    the list; a second loss term made of torch operations puts kernels into the graph which the list does not know: the graph stays the replay mechanism,
    and the losses still track the eager steps. """
    from deepcv_amd import modules, optim
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    batches = [(torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)) for _ in range(5)]
    curves = []
    for graph in (True, False):
        model = _model(params, 'image_classifier_model', device)
        opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
        losses, weights = _setup_losses({'ce': modules.CrossEntropyLoss(), 'l2': lambda p, y: (p * p).mean()}, {'ce': 1.0, 'l2': 0.1}, device)
        step = GraphedTrainStep(model, opt, losses, weights)
        got = []
        for xb, yb in batches:
            out, _ = step(xb, yb) if graph else step._eager(xb, yb)
            got.append(out['main_loss'].item())
        torch.cuda.synchronize()
        if graph:
            assert step._launch_list is None and step.graph_fb is not None
            assert step._list_verified == step._replay_list           # (DCV_REPLAY_LIST=0: no list was recorded, nothing to verify)
        curves.append(got)
    assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(*curves)), curves


def test_forward_launch_stages_the_batch_of_a_replayed_step(device, params):
    """ A captured step on the resident path has no `dcv_indirect_copy` launch: the forward launch reads the batch through the mailbox (and copies labels and
    optimizer scalars).  Batches the mailbox cannot point at (another memory layout: ordinary copy into the static tensor, job 0 then names that tensor)
    and ones it can alternate; with `DCV_STAGE_IN_FORWARD` off the same steps go through the staging kernel: same losses, bit for bit. """
    from deepcv_amd import modules, optim, resident
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    batches = [(torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)) for _ in range(6)]
    batches[3] = (batches[3][0].to(memory_format=torch.channels_last), batches[3][1])
    curves, staged = [], []
    for fold in (None, False):
        model = _model(params, 'image_classifier_model', device)
        opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
        losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
        step = GraphedTrainStep(model, opt, losses, weights)
        step._stage_in_forward = fold
        got = []
        for xb, yb in batches:
            out, _ = step(xb, yb)
            got.append(out['main_loss'].item())
        torch.cuda.synchronize()
        resident.check_status()
        staged.append(step._stage_in_forward)
        curves.append(got)
        assert torch.equal(step.static_x, batches[-1][0]) and torch.equal(step.static_y, batches[-1][1])      # the static copies the backward pass reads
    assert staged == [True, False]
    assert curves[0] == curves[1], curves


def test_hipgraph_replay_of_the_resident_step_is_not_a_slow_path(device, params):
    """ The launch-list replay falls back to replaying the captured hipGraph (`DCV_REPLAY_LIST=0`, a step with foreign kernels, a failed verification): same
    losses, and a comparable step time — a kernel's arguments and the mailbox sit in HOST memory for a graph node, and a system-scope atomic load of the
    mailbox word by every wave once made the forward launch 700 us there (0.2 -> 0.9 ms per step) while the list replay, which every measurement used, did
    not notice.  The bound is loose (3x): this is a tripwire, not a benchmark. """
    import time
    from deepcv_amd import modules, optim, resident
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    batches = [(torch.randn(512, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (512,), generator=g).to(device)) for _ in range(4)]
    curves, times = [], []
    for replay_list in (True, False):
        model = _model(params, 'image_classifier_model', device)
        opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
        losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
        step = GraphedTrainStep(model, opt, losses, weights)
        step._replay_list = replay_list
        got = []
        for i in range(8):                                   # eager step, capture, verification, first replays
            out, _ = step(*batches[i % 4])
            got.append(out['main_loss'].item())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            step(*batches[i % 4])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 200)
        resident.check_status()
        curves.append(got)
    assert curves[0] == curves[1], curves
    assert times[1] < 3.0 * times[0], f'hipGraph replay {times[1] * 1e3:.3f} ms/step against {times[0] * 1e3:.3f} ms/step of the launch-list replay'


@pytest.mark.parametrize('which', ['forward', 'backward'])
def test_a_block_which_never_arrives_is_reported_not_waited_for(device, params, which):
    """ The all-reduce's spins are bounded: with one block muted (test hook in the workspace's control words: it takes part in everything but never publishes
    its partial sums) every block gives up after its poll budget, the launch ends, the error word names the op, the step's loss is NaN and
    `resident.check_status()` raises — and the next step, with the hook cleared, is an ordinary step again. """
    from deepcv_amd import modules, resident
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)
    model = _model(params, 'image_classifier_model', device)
    ref = _model(params, 'image_classifier_model', device)
    prev, prev_b = resident.set_enabled(True), resident.set_backward_enabled(True)
    try:
        model.train()
        loss = modules.CrossEntropyLoss()(model(x), y)           # creates both workspaces
        loss.backward()
        torch.cuda.synchronize()
        resident.check_status()
        model.zero_grad(set_to_none=True)
        ws = (resident._ws if which == 'forward' else resident._ws_bwd)[device]
        ws[8:12].copy_(torch.tensor([4], dtype=torch.int32).view(torch.uint8).to(device))        # block 3 never publishes
        loss = modules.CrossEntropyLoss()(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        if which == 'forward':
            assert not torch.isfinite(loss).item()
        with pytest.raises(RuntimeError, match='timed out'):
            resident.check_status()
        assert int(ws[8:12].view(torch.int32).item()) == 0      # check_status wiped the workspace (hook included)
        # recovery: same weights as a fresh model (the failed steps never reached an optimizer), same batch -> same loss and gradients as the reference
        for m in (model, ref):
            m.zero_grad(set_to_none=True)
        for (n1, b1), (_, b2) in zip(model.named_buffers(), ref.named_buffers()):
            b1.copy_(b2)
        la, loss_a, _ = _run(model, x, y, True, True)
        lb, loss_b, _ = _run(ref, x, y, True, True)
        assert loss_a == loss_b and torch.equal(la, lb)
        for (name, pa), (_, pb) in zip(model.named_parameters(), ref.named_parameters()):
            assert torch.equal(pa.grad, pb.grad), name
    finally:
        resident.set_enabled(prev)
        resident.set_backward_enabled(prev_b)


def test_persistent_launches_gradients_match_oracle_at_the_bench_batch(device, params):
    """ Every parameter gradient of the stock spec's smooth (Tanh) twin at bs 512, computed by the two persistent launches, against the CPU oracle at the strict
    bound of tests/test_model_gpu.py (2e-3 of the tensor's largest gradient + 1e-5 of the model's): pins resident_bwd_kernel directly — weight gradients, head,
    in-launch cross-entropy path excluded (eager loss) — not through the per-layer kernels. """
    from test_model_gpu import _build, _batches, _smooth_twin
    from deepcv_amd import modules, resident
    native, oracle = _build({'twin': _smooth_twin(params['image_classifier_model'])}, 'twin', 10)
    native = native.to(device)
    (x, y), = _batches(1, 512)
    oracle.train()
    loss_o = torch.nn.CrossEntropyLoss()(oracle(x), y)
    loss_o.backward()
    prev, prev_b = resident.set_enabled(True), resident.set_backward_enabled(True)
    prev_cap, resident.capture_programs = resident.capture_programs, True
    try:
        before = (resident.launches(), resident.backward_launches())
        native.train()
        loss_n = modules.CrossEntropyLoss()(native(x.to(device)), y.to(device))
        loss_n.backward()
        torch.cuda.synchronize()
        resident.check_status()
        assert (resident.launches(), resident.backward_launches()) == (before[0] + 1, before[1] + 1), 'the persistent launches did not run'
        prog = resident.last_programs['backward'][0]
        assert prog.count == 5 and all(prog.op[i].dw_part for i in range(5)) and prog.head.enabled and prog.head.fc_dw, 'weight gradients / head are not inside the launch'
        assert not any(prog.op[i].dz for i in range(5)), 'dz must not leave the launch when nothing reads it'
    finally:
        resident.capture_programs = prev_cap
        resident.set_enabled(prev)
        resident.set_backward_enabled(prev_b)
    assert abs(loss_n.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item())
    gscale = max(p.grad.abs().max().item() for p in oracle.parameters())
    for (name, p_n), (_, p_o) in zip(native.named_parameters(), oracle.named_parameters()):
        abs_err = (p_n.grad.detach().cpu() - p_o.grad).abs().max().item()
        bound = 2e-3 * p_o.grad.abs().max().item() + 1e-5 * gscale
        assert abs_err <= bound, f'{name}: gradient error {abs_err:.3e} > {bound:.3e}'


def test_backward_launch_timeout_is_seen_by_the_training_loop(device, params):
    """ A backward launch which gives up (one block never publishes) has already published the step's loss: the failure must reach the host on the path a
    training loop reads anyway.  The launch poisons every gradient it owes with NaN, the optimizer carries it into the weights, and the NEXT loss read through
    `GraphedTrainStep.loss_values()` is NaN -> `resident.check_status()` -> RuntimeError naming the timed-out launch; nobody calls check_status here. """
    from deepcv_amd import modules, optim, resident
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    x, y = torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)
    model = _model(params, 'image_classifier_model', device)
    opt = optim.AdamW(model.parameters(), lr=1e-3)
    losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
    step = GraphedTrainStep(model, opt, losses, weights)
    prev, prev_b = resident.set_enabled(True), resident.set_backward_enabled(True)
    try:
        for _ in range(4):                                       # eager, capture, verified replay, replay
            step(x, y)
            assert all(v == v for v in step.loss_values().values())
        ws = resident._ws_bwd[device]
        ws[8:12].copy_(torch.tensor([4], dtype=torch.int32).view(torch.uint8).to(device))        # block 3 of the backward launch never publishes
        with pytest.raises(RuntimeError, match='timed out'):
            for _ in range(3):                                   # the failing step's own loss is still finite; the next one's is not
                step(x, y)
                step.loss_values()
        torch.cuda.synchronize()
    finally:
        if device in resident._ws_bwd:
            resident._ws_bwd[device][8:12].zero_()
        resident.set_enabled(prev)
        resident.set_backward_enabled(prev_b)


def test_launch_list_with_a_weight_gradient_batch_call_stays_in_bounds(device, params):
    """ `dcv_conv2d_wgrad_batch` writes a host OUT array (the slab segments of its jobs).  A launch list re-issues the call with the ctypes objects recorded at
    capture: the segment count must start from zero on every call (it used to append: a host heap overflow growing with every replay), and the replays must
    reproduce the eager steps bit for bit.  Weight gradients kept outside the persistent launch put the call into the captured step. """
    from deepcv_amd import modules, optim, resident
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    g = torch.Generator().manual_seed(DATA_SEED)
    batches = [(torch.randn(64, 3, 32, 32, generator=g).to(device), torch.randint(0, 10, (64,), generator=g).to(device)) for _ in range(7)]
    prev, resident._wgrad_in_launch = resident._wgrad_in_launch, False
    try:
        curves = []
        for graph in (True, False):
            model = _model(params, 'image_classifier_model', device)
            opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
            losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
            step = GraphedTrainStep(model, opt, losses, weights)
            got = []
            for xb, yb in batches:
                out, _ = step(xb, yb) if graph else step._eager(xb, yb)
                got.append(out['main_loss'].item())
            torch.cuda.synchronize()
            if graph and not step._replay_list:
                assert step._launch_list is None and step.graph_fb is not None            # DCV_REPLAY_LIST=0: hipGraph replays, no list to inspect
            elif graph:
                assert step._launch_list is not None, 'the step was expected to replay as a launch list'
                calls = [(fn, args) for fn, args, _ in step._launch_list.calls if getattr(fn, '__name__', '') == 'dcv_conv2d_wgrad_batch' or 'wgrad_batch' in repr(fn)]
                assert calls, [getattr(fn, '__name__', repr(fn)) for fn, _, _ in step._launch_list.calls]
                for _, args in calls:
                    n_jobs, ns = args[0], args[3]._obj
                    assert 0 < ns.value <= 2 * n_jobs, f'segment count {ns.value} after {len(batches) - 2} replays of {n_jobs} jobs'
            curves.append(got)
        assert curves[0] == curves[1], curves
    finally:
        resident._wgrad_in_launch = prev
