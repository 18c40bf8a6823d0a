""" CPU tests of the host-side mirror of the reference interface: YAML grammar, creator registry / argument injection, module construction, parameter
initialisation parity with the oracle, shape inference without kernels, error behaviour.  No GPU, no kernel launch. """
import copy

import pytest
import torch

from deepcv_amd import modules as native
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta import nn as deepcv_nn
from deepcv_amd.meta import nn_spec, submodule_creators
from deepcv_amd.meta.base_module import DeepcvModule
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.hyperparams import Hyperparameters, to_hyperparameters
from oracle import deepcv_oracle as O

SEED = 563454


def _pair(params, name, nc=10, shape=(3, 32, 32)):
    ds = {'trainset': SyntheticImageDataset(4, image_shape=shape, num_classes=nc)}
    torch.manual_seed(SEED)
    m = create_model(ds, params[name])
    torch.manual_seed(SEED)
    o = O.create_model(shape, params[name], nc)
    return m, o


@pytest.mark.parametrize('name,nc', [('image_classifier_model', 10), ('image_classifier_bn_only_model', 10), ('cifar100_deep_model', 100)])
def test_init_is_bit_identical_to_oracle(params, name, nc):
    m, o = _pair(params, name, nc)
    sd, sdo = m.state_dict(), o.state_dict()
    assert list(sd.keys()) == list(sdo.keys())
    for k in sd:
        assert sd[k].shape == sdo[k].shape and torch.equal(sd[k], sdo[k]), k


def test_stock_model_facts(params):
    """ 17 010 parameters, dense link output 20x8x8 -> 1280 features, per-layer capacities. """
    m, _ = _pair(params, 'image_classifier_model')
    assert deepcv_nn.get_model_capacity(m) == 17010
    backbone = m._submodules['_submodule_0']
    assert isinstance(backbone, DeepcvModule) and not backbone.is_sequential_nn() and m.is_sequential_nn()
    assert backbone._features_shapes[-1] == (20, 8, 8)
    assert m._features_shapes[-2] == (1280,) and m._features_shapes[-1] == (10,)
    assert backbone._submodules_capacities[:3] == [320, 420, 420] and backbone._submodules_capacities[4:6] == [656, 2384]
    assert 'pooling1' in backbone._submodules and backbone._submodule_references == {'_submodule_7': ['pooling1']}
    assert 'DeepcvModule' in str(m)


def test_native_modules_replace_torch_ones(params):
    m, _ = _pair(params, 'image_classifier_model')
    layer0 = m._submodules['_submodule_0']._submodules['_submodule_0']
    assert isinstance(layer0, deepcv_nn.LayerSequential)
    assert [type(c) for c in layer0] == [native.Conv2d, native.ReLU, native.BatchNorm2d, native.GroupNorm]   # Conv -> Act -> BN -> GN (nn.py:553, :511-512)
    assert isinstance(layer0[0], torch.nn.Conv2d) and deepcv_nn.is_conv(layer0[0])
    assert layer0[2].momentum == pytest.approx(0.07359778246238029) and layer0[3].num_groups == 4
    assert isinstance(m._submodules['_submodule_1'], native.Flatten)
    fc = m._submodules['_submodule_2']
    assert [type(c) for c in fc] == [native.Linear, native.Sigmoid] and fc[0].in_features == 1280
    assert isinstance(m._submodules['_submodule_0']._submodules['pooling1'], native.AvgPool2d)


def test_native_modules_refuse_cpu_compute(params):
    m, _ = _pair(params, 'image_classifier_model')
    with pytest.raises(RuntimeError, match='HIP'):
        m(torch.zeros(2, 3, 32, 32))
    from deepcv_amd import ops
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1, 1)


def test_shape_inference_runs_without_kernels():
    conv = native.Conv2d(3, 8, 5, stride=2, padding=2)
    assert tuple(deepcv_nn.get_out_features_shape((3, 33, 31), conv)) == tuple(torch.nn.Conv2d(3, 8, 5, stride=2, padding=2)(torch.zeros(1, 3, 33, 31)).shape)
    pool = native.MaxPool2d(3, 2, 1)
    assert tuple(deepcv_nn.get_out_features_shape((8, 17, 17), pool)) == (1, 8, 9, 9)
    assert tuple(deepcv_nn.get_out_features_shape((20, 8, 8), native.Flatten())) == (1, 1280)
    assert not conv.training   # the reference leaves the module in eval mode (nn.py:698)


def test_yaml_grammar_spellings():
    """ nn_spec.py:194-215: name | {name: params} | {name: [subm_name, params]} | {name: "subm_name"} | params with `_name`. """
    hp = {'act_fn': torch.nn.ReLU, 'architecture': [
        {'conv2d': {'kernel_size': 3, 'out_channels': 4}},
        {'conv2d': ['named', {'kernel_size': [3, 3], 'out_channels': 4}]},
        {'avg_pooling': 'pool_a'} if False else {'average_pooling': ['pool_a', {'kernel_size': 2}]},
        {'conv2d': {'_name': 'by_key', 'kernel_size': 1, 'out_channels': 2, 'act_fn': None}},
        torch.nn.Flatten,
        'torch.nn.Identity',
        {'torch.nn.Dropout': {'p': 0.0}},
        {'linear': {'out_features': 3, 'act_fn': torch.nn.Tanh}},
    ]}
    m = DeepcvModule((2, 8, 8), hp)
    assert list(m._submodules.keys()) == ['_submodule_0', 'named', 'pool_a', 'by_key', '_submodule_4', '_submodule_5', '_submodule_6', '_submodule_7']
    assert [type(c) for c in m._submodules['by_key']] == [native.Conv2d]               # local act_fn=None overrides the global ReLU
    assert m._submodules['_submodule_0'][0].padding == (1, 1) and m._submodules['by_key'][0].padding == (0, 0)   # default "same" padding (nn.py:393-399)
    assert isinstance(m._submodules['_submodule_4'], native.Flatten) and isinstance(m._submodules['_submodule_5'], torch.nn.Identity)
    assert m._submodules['_submodule_7'][0].in_features == 2 * 4 * 4 and isinstance(m._submodules['_submodule_7'][1], native.Tanh)
    assert m._features_shapes[-1] == (3,)


def test_global_local_hp_merge_and_nested_module():
    """ nn_spec.py:115-123: globals complete local params, local wins; nested modules inherit the parent's globals. """
    hp = {'act_fn': torch.nn.LeakyReLU, 'batch_norm': {'affine': True, 'eps': 1e-5, 'momentum': 0.1}, 'architecture': [
        {'_nested_deepcv_module': {'act_fn': torch.nn.ReLU, 'architecture': [{'conv2d': {'kernel_size': 3, 'out_channels': 4}},
                                                                               {'conv2d': {'kernel_size': 3, 'out_channels': 4, 'batch_norm': None}}]}},
        {'conv2d': {'kernel_size': 3, 'out_channels': 2}}]}
    m = DeepcvModule((3, 8, 8), hp)
    inner = m._submodules['_submodule_0']
    assert [type(c) for c in inner._submodules['_submodule_0']] == [native.Conv2d, native.ReLU, native.BatchNorm2d]
    assert [type(c) for c in inner._submodules['_submodule_1']] == [native.Conv2d, native.ReLU]
    assert [type(c) for c in m._submodules['_submodule_1']] == [native.Conv2d, native.LeakyReLU, native.BatchNorm2d]


def test_creator_registry_and_argument_injection():
    """ submodule_creators.py:133-160 + nn_spec.py:172-182: a custom creator receives exactly the arguments its signature names. """
    seen = {}

    def my_block(submodule_params, input_shape, prev_shapes, act_fn, width: int = 3):
        seen.update(submodule_params=submodule_params, input_shape=input_shape, n_prev=len(prev_shapes), act_fn=act_fn, width=width)
        return torch.nn.Identity()
    hp = {'act_fn': torch.nn.ReLU, 'width': 7, 'architecture': [{'conv2d': {'kernel_size': 1, 'out_channels': 5}}, {'my_block': {'extra': 1, 'act_fn': torch.nn.Tanh}}]}
    m = DeepcvModule((3, 4, 4), hp, additional_submodule_creators={'my_block': my_block})
    assert seen == {'submodule_params': {'extra': 1}, 'input_shape': (5, 4, 4), 'n_prev': 2, 'act_fn': torch.nn.Tanh, 'width': 7}
    assert isinstance(m._submodules['_submodule_1'], torch.nn.Identity)

    reg = {}

    @submodule_creators.submodule_creator_dec('thing', submodule_creators=reg, required_subm_params_keys={'k'})
    def thing(submodule_params):
        thing._check_submodule_params(submodule_params)
        return torch.nn.Identity()
    assert 'thing' in reg
    with pytest.raises(ValueError, match='Missing'):
        DeepcvModule((3, 4, 4), {'act_fn': None, 'architecture': [{'thing': {}}]}, additional_submodule_creators=reg)
    with pytest.raises(AssertionError):
        submodule_creators.submodule_creator_dec('conv2d')   # duplicates are refused (submodule_creators.py:137)
    assert {'conv2d', 'linear', 'fully_connected', 'average_pooling', 'avg_pooling', 'residual_link', 'dense_link'} <= set(submodule_creators.BASIC_SUBMODULE_CREATORS)


def test_spec_errors():
    base = {'act_fn': torch.nn.ReLU}
    with pytest.raises(ValueError, match='duplicate'):
        DeepcvModule((3, 8, 8), {**base, 'architecture': [{'conv2d': ['a', {'kernel_size': 3, 'out_channels': 2}]}, {'conv2d': ['a', {'kernel_size': 3, 'out_channels': 2}]}]})
    with pytest.raises(ValueError, match='reference'):
        DeepcvModule((3, 8, 8), {**base, 'architecture': [{'conv2d': {'kernel_size': 3, 'out_channels': 2}}, {'residual_link': {'_from': 'nope'}}]})
    with pytest.raises(RuntimeError, match='Could not locate'):
        DeepcvModule((3, 8, 8), {**base, 'architecture': [{'no_such_creator': {}}]})
    with pytest.raises(ValueError, match='Missing mandatory'):
        DeepcvModule((3, 8, 8), {'architecture': []})
    with pytest.raises(ValueError, match='_from'):
        DeepcvModule((3, 8, 8), {**base, 'architecture': [{'conv2d': {'kernel_size': 3, 'out_channels': 2}}, {'dense_link': {}}]})
    with pytest.raises(NotImplementedError):
        DeepcvModule((3, 8, 8), {**base, 'spectral_norm': {'name': 'weight'}, 'architecture': [{'conv2d': {'kernel_size': 3, 'out_channels': 2}}]})
    with pytest.raises(NotImplementedError):
        DeepcvModule((3, 8, 8), {**base, 'architecture': [{'_nas_layer_choice': {'_candidates': []}}]})


def test_links_shapes_and_reference_release(params):
    """ residual (sum) keeps the shape, dense (concat) adds channels, `allow_scaling` resizes (submodule_creators.py:272-327). """
    hp = {'act_fn': torch.nn.ReLU, 'architecture': [
        {'conv2d': ['c1', {'kernel_size': 3, 'out_channels': 4}]},
        {'conv2d': {'kernel_size': 3, 'out_channels': 4}},
        {'residual_link': ['r1', {'_from': 'c1'}]},
        {'average_pooling': {'kernel_size': 2}},
        {'dense_link': {'_from': ['c1', 'r1'], 'allow_scaling': True}}]}
    m = DeepcvModule((3, 8, 8), hp)
    assert m._features_shapes[3] == (4, 8, 8) and m._features_shapes[-1] == (12, 4, 4)
    assert m._submodule_references == {'r1': ['c1'], '_submodule_4': ['c1', 'r1']}
    hp_bad = copy.deepcopy(hp)
    hp_bad['architecture'][-1] = {'dense_link': {'_from': 'c1'}}
    with pytest.raises(RuntimeError, match='allow_scaling'):
        DeepcvModule((3, 8, 8), hp_bad)


def test_padding_and_gain_helpers():
    assert deepcv_nn.get_padding_from_kernel([5, 5]) == [2, 2] and deepcv_nn.get_padding_from_kernel(3) == 1 and deepcv_nn.get_padding_from_kernel([1, 4]) == [0, 1]
    assert deepcv_nn.get_gain_name(torch.nn.LeakyReLU) == 'leaky_relu' and deepcv_nn.get_gain_name(native.ReLU) == 'relu' and deepcv_nn.get_gain_name(torch.nn.GELU) == 'relu'
    assert deepcv_nn.is_conv(native.Conv2d) and deepcv_nn.is_fully_connected(native.Linear(2, 2)) and not deepcv_nn.is_conv(torch.nn.Linear)


def test_hyperparameters_container():
    hp, missing = to_hyperparameters({'a': 1}, defaults={'a': ..., 'b': 2, 'c': ...}, raise_if_missing=False)
    assert dict(hp) == {'a': 1, 'b': 2} and missing == ['c']
    with pytest.raises(ValueError):
        to_hyperparameters({}, defaults={'a': ...})
    assert isinstance(hp, Hyperparameters) and hp == {'a': 1, 'b': 2} and len(hp) == 2
    with pytest.raises(TypeError):
        hp['a'] = 3


def test_state_dict_roundtrip_with_reference_modules(params):
    """ Checkpoint compatibility: a state_dict of the reference's torch.nn composition loads into the native model and back. """
    m, o = _pair(params, 'image_classifier_model')
    with torch.no_grad():
        for p in o.parameters():
            p.add_(0.25)
    m.load_state_dict(o.state_dict())
    o2 = O.create_model((3, 32, 32), params['image_classifier_model'], 10)
    o2.load_state_dict(m.state_dict())
    for (k, a), (_, b) in zip(o.state_dict().items(), o2.state_dict().items()):
        assert torch.equal(a, b), k


def test_modules_without_hip_kernels_say_so_once(caplog):
    """ 1-D / 3-D operators stay on torch's kernels (outside every BASELINE config): warned once per type, not silent.  InstanceNorm2d, LayerNorm and
    LocalResponseNorm are native modules (GroupNorm statistics kernels with one channel per group; LayerNorm + its per-pixel affine, `dcv_layernorm_affine_*`;
    `dcv_lrn_*`): no warning; the factory hands LayerNorm the map's [H, W] as its normalised shape (reference `nn.py:492-493`). """
    import logging
    from deepcv_amd.meta import nn as dnn
    dnn._WARNED_NON_NATIVE.clear()
    with caplog.at_level(logging.WARNING):
        mods = dnn.normalization_techniques(input_shape=torch.Size((8, 4, 4)), layer_norm={'eps': 1e-5}, instance_norm={'eps': 1e-5})
        lrn = dnn.normalization_techniques_impl('local_response_norm', {'size': 3})[0]
        dnn.conv_nd(1, 3, 4, 3)
        dnn.conv_nd(1, 3, 4, 3)
        dnn.conv_nd(3, 3, 4, 3)
    assert [type(m).__name__ for m in mods] == ['LayerNorm', 'InstanceNorm2d']
    from deepcv_amd import modules
    assert isinstance(mods[0], modules.LayerNorm) and tuple(mods[0].normalized_shape) == (4, 4) and mods[0].weight.shape == (4, 4)
    assert isinstance(mods[1], modules.InstanceNorm2d) and isinstance(lrn, modules.LocalResponseNorm)
    assert modules.native_type(torch.nn.LayerNorm) is modules.LayerNorm and modules.native_type(torch.nn.LocalResponseNorm) is modules.LocalResponseNorm
    msgs = [r.getMessage() for r in caplog.records if 'no hand-written HIP kernel' in r.getMessage()]
    assert len(msgs) == 2 and any('Conv3d' in m for m in msgs) and any('Conv1d' in m for m in msgs) and not any('Norm' in m for m in msgs)


def test_resident_recording_is_refused_for_models_with_foreign_leaf_modules(params):
    """ `deepcv_amd.resident` records a forward only for models made of the native module kinds a resident program consists of: a Dropout (or any other
    module computing through its own kernels) would run on not-yet-filled tensors during a recording and advance its random-number generator on the way. """
    import copy
    from deepcv_amd import resident
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    ds = {'trainset': SyntheticImageDataset(8)}
    assert resident._foreign_module(create_model(ds, params['image_classifier_model'])) is None
    assert resident._foreign_module(create_model(ds, params['image_classifier_bn_only_model'])) is None
    with_dropout = copy.deepcopy(dict(params['image_classifier_model']))
    with_dropout['dropout_prob'] = 0.3
    assert 'Dropout' in resident._foreign_module(create_model(ds, with_dropout))
    wide = create_model({'trainset': SyntheticImageDataset(8, image_shape=(3, 224, 224), num_classes=1000)}, params['resnet18_style_model'])
    assert 'MaxPool2d' in resident._foreign_module(wide)


def test_resident_recording_state_is_per_thread():
    """ A forward is recorded by the thread which runs it (deepcv_amd/resident.py): an op issued meanwhile by another thread must not see the recorder — neither
    be appended to its program nor raise `Ineligible` outside any `DeepcvModule.forward`. """
    import threading
    import torch
    from deepcv_amd import resident
    seen = {}
    with resident.Recorder(torch.zeros(2, 3, 8, 8)) as rec:
        assert resident.recorder() is rec
        with pytest.raises(resident.Ineligible):
            resident.forbid('an op of the recording thread')

        def other():
            seen['recorder'] = resident.recorder()
            try:
                resident.forbid('an op of another thread')
                seen['raised'] = False
            except resident.Ineligible:
                seen['raised'] = True
            resident.set_staging((1, 0, 2))
            seen['staging'] = resident.staging_consumed()
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert resident.staged_program() is None and not resident.staging_consumed()      # the other thread's staging request is its own
    assert seen == {'recorder': None, 'raised': False, 'staging': False}
    assert resident.recorder() is None


@pytest.mark.parametrize('shape,kind', [((64, 3, 3, 3), 'col'), ((64, 3, 7, 7), 'fold'), ((128, 1, 13, 13), 'fold4')])
def test_filter_regroup_index_maps_reproduce_the_torch_relayout_and_its_gradient(monkeypatch, shape, kind):
    """ `ops._regroup` replaces pad / reshape / permute on a first layer's filter by one gather launch each way; the index maps are host logic: forward map =
    the re-layout, inverse map = its autograd (every filter element occurs once, padding is zero).  The launch itself is stood in for by index_select here. """
    import torch.nn.functional as F
    from deepcv_amd import ops

    class Gather(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, fwd, inv, out_shape):
            ctx.inv, ctx.wshape = inv, w.shape
            flat = torch.cat([w.reshape(-1), w.new_zeros(1)])
            return flat[fwd.long()].reshape(out_shape)             # (index -1 -> the appended zero)

        @staticmethod
        def backward(ctx, g):
            flat = torch.cat([g.reshape(-1), g.new_zeros(1)])
            return flat[ctx.inv.long()].reshape(ctx.wshape), None, None, None
    monkeypatch.setattr(ops, '_RegroupFn', Gather)
    cout, cin, r, s = shape
    if kind == 'col':
        k, kpad = cin * r * s, 64
        relayout, key = (lambda w: F.pad(w.reshape(cout, k), (0, kpad - k)).reshape(cout, kpad, 1, 1)), ('col', kpad)
    else:
        sh = 2 if kind == 'fold' else 4
        a = -(-r // sh)

        def relayout(w):
            w = F.pad(w, (0, a * sh - s, 0, a * sh - r))
            w = w.reshape(cout, cin, a, sh, a, sh).permute(0, 3, 5, 1, 2, 4).reshape(cout, sh * sh * cin, a, a)
            return F.pad(w, (0, 0, 0, 0, 0, 16 - sh * sh * cin))
        key = ('fold', sh, sh, a)
    torch.manual_seed(3)
    w1 = torch.randn(shape, requires_grad=True)
    w2 = w1.detach().clone().requires_grad_(True)
    ref, got = relayout(w1), ops._regroup(w2, key, relayout)
    assert got.shape == ref.shape and torch.equal(got, ref)
    g = torch.randn_like(ref)
    ref.backward(g); got.backward(g)
    assert torch.equal(w1.grad, w2.grad)
