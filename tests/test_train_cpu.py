""" `train()` / engine control-plane tests on CPU.  The model here is the ORACLE's torch.nn composition (the native modules refuse CPU tensors by design):
what is under test is the host logic of the training driver — event order, per-iteration LR scheduling, loss bookkeeping, checkpoints, crash / resume,
returned tuple — against the semantics of `src/deepcv/meta/ignite_training.py:178-370`. """
import math
from pathlib import Path

import pytest
import torch

from deepcv_amd.meta import engine as E
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import BackendConfig, _batch_losses, _setup_losses, train
from oracle import deepcv_oracle as O

SEED = 563454


def _hp(params, tmp_path, **over):
    hp = dict(params['train_image_classifier'])
    hp.update(epochs=2, batch_size=16, output_path=tmp_path, num_workers=0, save_every_iters=3, log_progress_every_iters=2, prefetch_batches=False, hip_graph=False)
    hp.update(over)
    return hp


def _model(params):
    torch.manual_seed(SEED)
    return O.create_model((3, 32, 32), params['image_classifier_bn_only_model'], 10)


def test_engine_event_order_and_filters():
    log = []
    eng = E.Engine(lambda e, b: b * 2)
    eng.add_event_handler(E.Events.STARTED, lambda e: log.append('S'))
    eng.add_event_handler(E.Events.EPOCH_STARTED, lambda e: log.append(f'ES{e.state.epoch}'))
    eng.add_event_handler(E.Events.ITERATION_STARTED, lambda e: log.append(f'is{e.state.iteration}'))
    eng.add_event_handler(E.Events.ITERATION_COMPLETED(every=2), lambda e: log.append(f'ic{e.state.iteration}={e.state.output}'))
    eng.add_event_handler(E.Events.ITERATION_STARTED(once=3), lambda e: log.append('once3'))
    eng.add_event_handler(E.Events.EPOCH_COMPLETED, lambda e: log.append(f'EC{e.state.epoch}'))
    eng.add_event_handler(E.Events.COMPLETED, lambda e: log.append('C'))
    state = eng.run([1, 2, 3], max_epochs=2)
    assert log == ['S', 'ES1', 'is1', 'is2', 'ic2=4', 'is3', 'once3', 'EC1', 'ES2', 'is4', 'ic4=2', 'is5', 'is6', 'ic6=6', 'EC2', 'C']
    assert state.iteration == 6 and state.epoch == 2 and state.epoch_length == 3 and state.max_epochs == 2
    with pytest.raises(ValueError):
        E.Events.ITERATION_COMPLETED(every=2, once=1)


def test_metrics():
    acc = E.Accuracy(); acc.reset()
    acc.update((torch.tensor([[0.1, 0.9], [0.8, 0.2], [0.3, 0.7]]), torch.tensor([1, 0, 0])))
    assert acc.compute() == pytest.approx(2 / 3)
    loss = E.Loss(torch.nn.functional.mse_loss); loss.reset()
    loss.update((torch.tensor([1., 2.]), torch.tensor([0., 0.])))
    loss.update((torch.tensor([3.]), torch.tensor([0.])))
    assert loss.compute() == pytest.approx((2.5 * 2 + 9.0 * 1) / 3)
    eng = E.Engine(lambda e, b: {'l': float('nan') if b == 2 else 1.0})
    eng.add_event_handler(E.Events.ITERATION_COMPLETED, E.TerminateOnNan())
    assert eng.run([1, 2, 3], max_epochs=3).iteration == 2


def test_loss_setup_semantics():
    ce = torch.nn.CrossEntropyLoss()
    named, w = _setup_losses(ce, None, 'cpu')
    assert list(named) == ['main_loss'] and w['main_loss'] == 1.
    y_pred, y = torch.randn(4, 3), torch.tensor([0, 1, 2, 1])
    out = _batch_losses(named, w, y_pred, y)
    assert torch.equal(out['main_loss'], ce(y_pred, y))
    named, w = _setup_losses([ce, torch.nn.MultiMarginLoss()], [3., 1.], 'cpu')
    assert list(named) == ['loss_term_0', 'loss_term_1']
    out = _batch_losses(named, w, y_pred, y)
    expect = (3. * ce(y_pred, y) + torch.nn.MultiMarginLoss()(y_pred, y)) / 4.   # weighted MEAN (ignite_training.py:190-192)
    assert out['main_loss'].item() == pytest.approx(expect.item(), rel=1e-6)
    with pytest.raises(TypeError):
        _setup_losses({'a': ce}, [1.], 'cpu')


def test_train_matches_oracle_loop_on_cpu(params, tmp_path):
    """ One batch per epoch (so that shuffling cannot reorder batches): losses per iteration must equal the oracle's restatement of the same loop. """
    ds = SyntheticImageDataset(16, seed=3)
    hp = _hp(params, tmp_path, epochs=4, batch_size=16, validate_every_epochs=100)
    seen = []

    class Spy:
        def fire(self, event, **kw):
            if event == 'ON_ITERATION_COMPLETED':
                seen.append(kw['engine'].state.output['main_loss'])
    model = _model(params)
    metrics, state = train(hp, model, losses=torch.nn.CrossEntropyLoss(), datasets={'trainset': ds, 'validset': SyntheticImageDataset(8, seed=4)},
                           opt=torch.optim.AdamW, backend_conf=BackendConfig('cpu'), metrics={'accuracy': E.Accuracy()}, callbacks_handler=Spy())
    assert state.iteration == 4 and state.epoch == 4 and isinstance(state, E.State)
    ref_model = _model(params)
    ref = O.train_steps(ref_model, [(ds.images, ds.targets)], dict(hp), 4, iterations_per_epoch=1)
    assert seen == pytest.approx(ref, rel=2e-5)
    assert set(metrics) == {'valid_accuracy', 'valid_main_loss'} and 0. <= metrics['valid_accuracy'] <= 1. and math.isfinite(metrics['valid_main_loss'])


def test_checkpoints_crash_and_resume(params, tmp_path):
    ds = {'trainset': SyntheticImageDataset(64, seed=5), 'validset': SyntheticImageDataset(16, seed=6)}
    hp = _hp(params, tmp_path / 'a', crash_iteration=6)
    with pytest.raises(RuntimeError, match='STOP at iteration: 6'):
        train(hp, _model(params), loss=torch.nn.CrossEntropyLoss(), datasets=ds, opt=torch.optim.AdamW, backend_conf=BackendConfig('cpu'))   # `loss=` alias (Appendix A #24)
    ckpts = sorted((tmp_path / 'a').rglob('training_checkpoint_*.pt'))
    assert [p.name for p in ckpts] == ['training_checkpoint_3.pt']    # iteration 6 crashed before completing
    ckpt = torch.load(ckpts[0], weights_only=False)
    assert set(ckpt) == {'trainer', 'model', 'optimizer', 'scheduler'} and ckpt['trainer']['iteration'] == 3 and ckpt['scheduler'] == {'event_index': 3}
    assert set(ckpt['model']) == set(_model(params).state_dict())

    hp2 = _hp(params, tmp_path / 'b', resume_from=str(ckpts[0]))
    its = []

    class Spy:
        def fire(self, event, **kw):
            if event == 'ON_ITERATION_STARTED':
                its.append(kw['engine'].state.iteration)
    _, state = train(hp2, _model(params), losses=torch.nn.CrossEntropyLoss(), datasets=ds, opt=torch.optim.AdamW, backend_conf=BackendConfig('cpu'), callbacks_handler=Spy())
    assert its == [4, 5, 6, 7, 8] and state.iteration == 8 and state.epoch == 2   # resumes after the 3 iterations already done


def test_backend_config():
    bc = BackendConfig('cpu')
    assert bc.is_cpu and not bc.is_cuda and not bc.distributed and bc.rank == 0 and str(bc).startswith('single-node-cpu-')
    assert BackendConfig('cpu', dist_backend='gloo', dist_url='tcp://127.0.0.1:1').distributed
    assert not BackendConfig('cpu', dist_backend='gloo').distributed


def test_train_argument_validation(params, tmp_path):
    with pytest.raises(ValueError, match='mandatory'):
        train(_hp(params, tmp_path), _model(params), datasets={'trainset': SyntheticImageDataset(4)}, opt=torch.optim.AdamW)
    hp = _hp(params, tmp_path); hp.pop('epochs')
    with pytest.raises(ValueError, match='Missing mandatory'):
        train(hp, _model(params), losses=torch.nn.CrossEntropyLoss(), datasets={'trainset': SyntheticImageDataset(4), 'validset': SyntheticImageDataset(4)}, opt=torch.optim.AdamW,
              backend_conf=BackendConfig('cpu'))


def test_published_loss_wait_is_bounded():
    """ `GraphedTrainStep._wait_published` polls pinned memory for the sequence number the replayed graph publishes.  A replay which faults, or whose write
    never lands, must surface as an exception naming the expected and the seen sequence numbers — not as a host thread spinning forever. """
    import numpy as np
    from deepcv_amd.meta.ignite_training import GraphedTrainStep

    class _Stream:
        def __init__(self, drained, fail=False):
            self.drained, self.fail, self.queries = drained, fail, 0

        def query(self):
            self.queries += 1
            if self.fail:
                raise RuntimeError('HIP error: an illegal memory access was encountered')
            return self.drained

    def step(pub, launched, stream, timeout=30.):
        s = GraphedTrainStep(model=None, optimizer=None, losses=None, weights=None)
        s._pub_seq, s._launched_seq, s._seq_seen, s._launch_stream, s.publish_timeout_s = np.array([pub], dtype='<i4'), launched, 0, stream, timeout
        return s

    s = step(7, 7, _Stream(False))                      # already published: returns without querying anything
    s._wait_published()
    assert s._seq_seen == 7 and s._launch_stream.queries == 0
    with pytest.raises(RuntimeError, match=r'replay #8 .* drained without publishing.* seen in pinned memory: 7'):
        step(7, 8, _Stream(True))._wait_published()     # the stream is idle and the number never arrived: lost write
    with pytest.raises(RuntimeError, match=r'replay #8 .*nothing published after .* seen in pinned memory: 3'):
        step(3, 8, _Stream(False), timeout=0.05)._wait_published()
    with pytest.raises(RuntimeError, match='illegal memory access'):
        step(3, 8, _Stream(False, fail=True))._wait_published()   # an asynchronous device error surfaces through the stream query
