""" Generates the committed golden vectors from the CPU oracle (torch CPU ops composed like the reference; `oracle/`).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference itself has no numeric fixtures and cannot be imported, so these vectors pin THIS repository's restatement of its intended
semantics: they catch drift of the oracle (torch version, edits) on CPU, and they are what the HIP path is compared against on a GPU box where
`/root/reference` does not exist.  Inputs are regenerated from seeds by the tests; only expected outputs (and tiny inputs) are stored. """
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from deepcv_amd.utils import load_parameters  # noqa: E402  (YAML loader only)
from oracle import deepcv_oracle as O  # noqa: E402
from oracle import ops_ref  # noqa: E402

OUT = Path(__file__).resolve().parent
MODEL_SEED, DATA_SEED = 563454, 434546


def batches(n, bs, num_classes=10):
    g = torch.Generator().manual_seed(DATA_SEED)
    return [(torch.randn(bs, 3, 32, 32, generator=g), torch.randint(0, num_classes, (bs,), generator=g)) for _ in range(n)]


def state_hash(model) -> str:
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode()); h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def model_fixture(params, spec):
    torch.manual_seed(MODEL_SEED)
    m = O.create_model((3, 32, 32), params[spec], 10)
    out = {'init_sha256': np.array(state_hash(m)), 'init_first_conv_weight': next(m.parameters()).detach().numpy().copy()}
    (x, y), = batches(1, 8)
    m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    out['logits'], out['loss'] = logits.detach().numpy(), np.array(loss.item())
    names = [n for n, _ in m.named_parameters()]
    picks = [names[0], names[-2], names[-1]] + [n for n in names if n.endswith('_submodule_5.2.weight') or n.endswith('_submodule_4.0.weight')]
    for n, p in m.named_parameters():
        if n in picks:
            out['grad::' + n] = p.grad.detach().numpy().copy()
    # loss curve: 6 steps of the stock training recipe (AdamW + piecewise-linear LR over 6 iterations), batch 16
    torch.manual_seed(MODEL_SEED)
    m2 = O.create_model((3, 32, 32), params[spec], 10)
    hp = dict(params['train_image_classifier']); hp['epochs'] = 1
    out['train_losses'] = np.array(O.train_steps(m2, batches(6, 16), hp, 6, iterations_per_epoch=6))
    out['trained_last_fc_weight'] = dict(m2.named_parameters())[names[-2]].detach().numpy().copy()
    # data-parallel emulation, 2 and 4 shards
    for world in (2, 4):
        torch.manual_seed(MODEL_SEED)
        m3 = O.create_model((3, 32, 32), params[spec], 10)
        out[f'dp{world}_losses'] = np.array(O.train_steps_data_parallel(m3, batches(3, 16), hp, 3, world, iterations_per_epoch=3))
    return out


def ops_fixture():
    g = torch.Generator().manual_seed(2024)
    out = {}
    for name, (n, cin, h, w, cout, k, stride, pad, act) in {'conv_stock_5x5': (2, 4, 32, 32, 4, 5, 1, 2, 1), 'conv_stock_3x3': (2, 16, 16, 16, 16, 3, 1, 1, 2),
                                                            'conv_wide_3x3': (1, 32, 8, 8, 64, 3, 1, 1, 1), 'conv_1x1_s2': (1, 32, 8, 8, 64, 1, 2, 0, 0)}.items():
        x, wt, b = torch.randn(n, cin, h, w, generator=g), torch.randn(cout, cin, k, k, generator=g) * 0.1, torch.randn(cout, generator=g)
        ho = (h + 2 * pad - k) // stride + 1
        dy = torch.randn(n, cout, ho, ho, generator=g)
        r = ops_ref.conv2d_act(x, wt, b, stride, pad, 1, act, 0.01, dy)
        out.update({f'{name}::x': x.numpy(), f'{name}::w': wt.numpy(), f'{name}::b': b.numpy(), f'{name}::dy': dy.numpy(),
                    f'{name}::cfg': np.array([stride, pad, act]), **{f'{name}::{k_}': v.numpy() for k_, v in r.items()}})
    for c in (4, 16, 64):
        x, gam, bet, dy = torch.randn(4, c, 8, 8, generator=g) * 2 + 0.5, torch.rand(c, generator=g), torch.randn(c, generator=g), torch.randn(4, c, 8, 8, generator=g)
        r = ops_ref.batch_norm_train(x, gam, bet, torch.zeros(c), torch.ones(c), 0.07359778246238029, 1e-5, dy)
        out.update({f'bn{c}::x': x.numpy(), f'bn{c}::gamma': gam.numpy(), f'bn{c}::beta': bet.numpy(), f'bn{c}::dy': dy.numpy(), **{f'bn{c}::{k_}': v.numpy() for k_, v in r.items()}})
    x, gam, bet, dy = torch.randn(4, 16, 8, 8, generator=g), torch.rand(16, generator=g) + .5, torch.randn(16, generator=g), torch.randn(4, 16, 8, 8, generator=g)
    r = ops_ref.group_norm(x, 4, gam, bet, 1e-5, dy)
    out.update({'gn::x': x.numpy(), 'gn::gamma': gam.numpy(), 'gn::beta': bet.numpy(), 'gn::dy': dy.numpy(), **{f'gn::{k_}': v.numpy() for k_, v in r.items()}})
    x, dy = torch.randn(2, 8, 8, 8, generator=g), torch.randn(2, 8, 4, 4, generator=g)
    for mode in ('avg', 'max'):
        r = ops_ref.pool2d(x, 2, 2, 0, mode, dy)
        out.update({f'pool_{mode}::y': r['y'].numpy(), f'pool_{mode}::dx': r['dx'].numpy()})
    out.update({'pool::x': x.numpy(), 'pool::dy': dy.numpy()})
    r = ops_ref.interpolate_bilinear(x, (4, 4), False, dy)
    out.update({'interp::y': r['y'].numpy(), 'interp::dx': r['dx'].numpy()})
    x, wt, b, dy = torch.randn(8, 1280, generator=g), torch.randn(10, 1280, generator=g) * 0.03, torch.randn(10, generator=g), torch.randn(8, 10, generator=g)
    r = ops_ref.linear_act(x, wt, b, 3, 0., dy)
    out.update({'linear::x': x.numpy(), 'linear::w': wt.numpy(), 'linear::b': b.numpy(), 'linear::dy': dy.numpy(), **{f'linear::{k_}': v.numpy() for k_, v in r.items()}})
    z, t = torch.sigmoid(torch.randn(16, 10, generator=g)), torch.randint(0, 10, (16,), generator=g)     # CE on top of sigmoid outputs, like the stock spec
    r = ops_ref.cross_entropy(z, t)
    out.update({'ce::logits': z.numpy(), 'ce::target': t.numpy(), 'ce::loss': r['loss'].numpy(), 'ce::dlogits': r['dlogits'].numpy()})
    ps = [torch.randn(5, 7, generator=g), torch.randn(11, generator=g)]
    gs = [[torch.randn(5, 7, generator=g), torch.randn(11, generator=g)] for _ in range(3)]
    out.update({'opt::p0': ps[0].numpy(), 'opt::p1': ps[1].numpy(), **{f'opt::g{t_}_{i}': gs[t_][i].numpy() for t_ in range(3) for i in range(2)}})
    for name, cls, kw in (('adamw', torch.optim.AdamW, dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False)),
                          ('sgd_nesterov', torch.optim.SGD, dict(lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)),
                          ('sgd_momentum', torch.optim.SGD, dict(lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=False))):
        res = ops_ref.optimizer_steps(cls, kw, ps, gs)
        out.update({f'opt::{name}_p0': res[0].numpy(), f'opt::{name}_p1': res[1].numpy()})
    hp = {'epochs': 2, 'optimizer_opts': {'lr': 1e-3}}
    out['sched::piecewise'] = np.array([O.piecewise_linear(O.stock_milestones(hp, 10), t_) for t_ in range(22)])
    out['sched::one_cycle'] = np.array(O.one_cycle_values(20, 1e-4, 0.1, 1e-4, 1e-2))
    return out


if __name__ == '__main__':
    params = load_parameters()
    np.savez_compressed(OUT / 'stock_model.npz', **model_fixture(params, 'image_classifier_model'))
    np.savez_compressed(OUT / 'bn_only_model.npz', **model_fixture(params, 'image_classifier_bn_only_model'))
    np.savez_compressed(OUT / 'ops.npz', **ops_fixture())
    for f in sorted(OUT.glob('*.npz')):
        print(f.name, f.stat().st_size, 'bytes')
