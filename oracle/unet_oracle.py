"""Oracle: the reference 2D U-Net forward as a flat list of functional torch-CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows common/model/unet.py of the reference:
  * conv unit  = conv3x3(pad 1)+bias -> Dropout2d -> BatchNorm2d(eval) -> ReLU   (unet.py:8-23)
  * block      = two conv units; which unit carries dropout: unet.py:63-82
  * down level = block, keep skip, max-pool 2                                      (unet.py:85-95)
  * up level   = nearest x2 -> conv3x3 (bias only) -> centre pad -> cat((up, skip)) -> block
                                                                                 (unet.py:98-120)
  * heads      = conv unit + conv1x1 for logits, optional twin for sigma          (unet.py:160-186)
Dropout is never sampled here: the per-(sample, channel) factors {0, 1/(1-p)} are an input
(``masks[site]`` of shape [N, C_site]); ``masks=None`` is eval mode (all Dropout2d inactive), which
is what set_dropout_mode(model, False) gives (common/utils/torchhelper.py:44-50).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, used unchanged by unet.py:17


def _dropout_rule(dropout_center, level, depth, is_down):
    """Which conv unit(s) of a block get a Dropout2d (unet.py:74-82)."""
    if dropout_center is None:
        return 'all'
    if level == depth:
        return 'no'
    if level + dropout_center >= depth:
        return 'last' if is_down else 'first'
    return 'no'


def _unit_has_dropout(dropout, rule, i, reps=2):
    """unet.py:63-72."""
    if dropout is None:
        return False
    return rule == 'all' or (rule == 'first' and i == 0) or (rule == 'last' and i == reps - 1)


def unet_plan(nb_classes, in_channels, depth=4, start_filters=16, dropout=0.2, dropout_center=None,
              residual=False, sigma_out=False, provide_features=False, bn=True):
    """Flat execution plan.  Each entry: dict(kind, key, cin, cout, site) where ``key`` is the
    state_dict prefix of the conv unit and ``site`` the index of its Dropout2d in execution order
    (= torch ``named_modules`` order) or None.  Defaults are the reference's (unet.py:124-130)."""
    plan = []
    sites = []

    def unit(key, cin, cout, has_do, relu=True):
        site = None
        if has_do:
            site = len(sites)
            sites.append((key + '.conv2d_batch_relu.dropout', cout))
        plan.append(dict(kind='unit', key=key + '.conv2d_batch_relu', cin=cin, cout=cout, site=site, bn=bn, relu=relu))

    def block(prefix, cin, cout, rule):
        # ConvBlock (unet.py:26-39) or, with ``residual``, ConvResidualBlock (unet.py:42-60): the last unit has no ReLU and the
        # block input goes through a 1x1 conv that is ADDED to the block output (nothing follows the sum)
        if residual:
            plan.append(dict(kind='res_begin'))
        for i in range(2):
            unit('{}.{}'.format(prefix, i), cin if i == 0 else cout, cout, _unit_has_dropout(dropout, rule, i),
                 relu=not (residual and i == 1))
        if residual:
            plan.append(dict(kind='res_add', key=prefix[:-len('.block')] + '.residual', cin=cin, cout=cout))

    cin, cout = in_channels, start_filters
    for lvl in range(depth):
        block('down_convs.{}.block.block'.format(lvl), cin, cout, _dropout_rule(dropout_center, lvl, depth, True))
        plan.append(dict(kind='pool'))
        cin, cout = cout, cout * 2
    block('bottom_convs.block', cin, cout, _dropout_rule(dropout_center, depth, depth, True))
    for j, lvl in enumerate(range(depth - 1, -1, -1)):
        cin = cout
        cout = cin // 2
        plan.append(dict(kind='up', key='up_convs.{}.upconv.1'.format(j), cin=cin, cout=cout))
        block('up_convs.{}.block.block'.format(j), 2 * cout, cout, _dropout_rule(dropout_center, lvl, depth, False))
    plan.append(dict(kind='fork'))
    unit('conv_cls.0', cout, cout, dropout is not None)
    plan.append(dict(kind='head', key='conv_cls.1', cin=cout, cout=nb_classes, out='logits'))
    if sigma_out:
        plan.append(dict(kind='rewind'))
        unit('conv_sigma.0', cout, cout, dropout is not None)
        plan.append(dict(kind='head', key='conv_sigma.1', cin=cout, cout=nb_classes, out='sigma'))
    return plan, sites


def strip_module_prefix(state):
    """Checkpoints saved from nn.DataParallel carry a 'module.' prefix (context.py:167, 176-179)."""
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state.items()}


def unet_forward(state, x, masks=None, return_features=False, **params):
    """x: float32 [N, Cin, H, W] -> logits [N, C, H, W] or (logits, sigma) when sigma_out; with
    ``return_features`` also the input of conv_cls (UNet.features, unet.py:178-179) as the last element.
    ``state``: mapping name -> tensor/ndarray with the reference's state_dict keys.
    ``masks``: None or a list (one per dropout site, execution order) of [N, C_site] factors."""
    state = {k: torch.as_tensor(v) for k, v in strip_module_prefix(state).items()}
    plan, sites = unet_plan(**params)
    if masks is not None and len(masks) != len(sites):
        raise ValueError('expected {} dropout masks, got {}'.format(len(sites), len(masks)))
    x = torch.as_tensor(x, dtype=torch.float32)
    skips = []
    outs = {}
    trunk = None
    for op in plan:
        kind = op['kind']
        if kind == 'unit':
            k = op['key']
            x = F.conv2d(x, state[k + '.conv.weight'], state[k + '.conv.bias'], padding=1)
            if op['site'] is not None and masks is not None:
                m = torch.as_tensor(masks[op['site']], dtype=torch.float32)
                x = x * m[:, :, None, None]
            if op['bn']:
                x = F.batch_norm(x, state[k + '.bn.running_mean'], state[k + '.bn.running_var'],
                                 state[k + '.bn.weight'], state[k + '.bn.bias'], False, 0.0, BN_EPS)
            if op.get('relu', True):
                x = F.relu(x)
        elif kind == 'res_begin':
            block_in = x
        elif kind == 'res_add':
            x = x + F.conv2d(block_in, state[op['key'] + '.weight'], state[op['key'] + '.bias'])
        elif kind == 'pool':
            skips.append(x)
            x = F.max_pool2d(x, 2)
        elif kind == 'up':
            skip = skips.pop()
            up = F.interpolate(x, scale_factor=2, mode='nearest')
            up = F.conv2d(up, state[op['key'] + '.weight'], state[op['key'] + '.bias'], padding=1)
            if tuple(up.shape[-2:]) < tuple(skip.shape[-2:]):  # unet.py:110-116
                dh = skip.shape[-2] - up.shape[-2]
                dw = skip.shape[-1] - up.shape[-1]
                up = F.pad(up, (dw // 2, dw // 2 + dw % 2, dh // 2, dh // 2 + dh % 2))
            x = torch.cat((up, skip), 1)
        elif kind == 'fork':
            trunk = x
        elif kind == 'rewind':
            x = trunk
        elif kind == 'head':
            outs[op['out']] = F.conv2d(x, state[op['key'] + '.weight'], state[op['key'] + '.bias'])
    result = (outs['logits'], outs['sigma']) if 'sigma' in outs else (outs['logits'],)
    if return_features:
        result = result + (trunk,)
    return result if len(result) > 1 else result[0]


def postnet_forward(state, x, nb_convs=3, masks=None):
    """common/model/postnet.py:6-18: nb_convs x [Conv2d 1x1 C->C, Dropout2d, BatchNorm2d (eval), ReLU] + Conv2d 1x1.  ``masks``:
    None (dropout None as in every shipped config, or Dropout2d in eval mode) or one [N, C] factor array per conv (MC-dropout)."""
    state = {k: torch.as_tensor(v) for k, v in strip_module_prefix(state).items()}
    x = torch.as_tensor(x, dtype=torch.float32)
    for i in range(nb_convs):
        k = 'convs.{}.conv2d_batch_relu'.format(i)
        x = F.conv2d(x, state[k + '.conv.weight'], state[k + '.conv.bias'])
        if masks is not None:
            x = x * torch.as_tensor(masks[i], dtype=torch.float32)[:, :, None, None]
        x = F.batch_norm(x, state[k + '.bn.running_mean'], state[k + '.bn.running_var'], state[k + '.bn.weight'],
                         state[k + '.bn.bias'], False, 0.0, BN_EPS)
        x = F.relu(x)
    return F.conv2d(x, state['conv_logits.weight'], state['conv_logits.bias'])


def postnet_synthetic_state(seed, in_channels, nb_classes, nb_convs=3):
    """Deterministic PostNet weights with the reference's keys (same recipe as synthetic_state)."""
    import math
    g = torch.Generator().manual_seed(seed)
    state = {}
    bound = 1.0 / math.sqrt(in_channels)
    for i in range(nb_convs):
        k = 'convs.{}.conv2d_batch_relu'.format(i)
        state[k + '.conv.weight'] = (torch.rand(in_channels, in_channels, 1, 1, generator=g) * 2 - 1) * bound
        state[k + '.conv.bias'] = (torch.rand(in_channels, generator=g) * 2 - 1) * bound
        state[k + '.bn.weight'] = torch.rand(in_channels, generator=g) * 0.5 + 0.75
        state[k + '.bn.bias'] = torch.randn(in_channels, generator=g) * 0.1
        state[k + '.bn.running_mean'] = torch.randn(in_channels, generator=g) * 0.1
        state[k + '.bn.running_var'] = torch.rand(in_channels, generator=g) + 0.5
        state[k + '.bn.num_batches_tracked'] = torch.tensor(1)
    state['conv_logits.weight'] = (torch.rand(nb_classes, in_channels, 1, 1, generator=g) * 2 - 1) * bound
    state['conv_logits.bias'] = (torch.rand(nb_classes, generator=g) * 2 - 1) * bound
    return state


def sample_masks(sites, n, p, generator):
    """Bernoulli(1-p) keep masks scaled by 1/(1-p), one [n, C] array per site -- what Dropout2d
    applies per (sample, channel) (torch feature_dropout)."""
    keep = 1.0 - p
    return [(torch.rand(n, c, generator=generator) < keep).float() / keep for _, c in sites]


def synthetic_state(seed, **params):
    """Deterministic random weights with the reference's state_dict keys/shapes and torch's default
    init distributions (kaiming-uniform(a=sqrt 5) conv weights, uniform bias), BN running stats
    randomised so that eval-mode BN is not the identity.  Pure torch-CPU generator
    draws: identical on every machine with the same torch build."""
    import math
    g = torch.Generator().manual_seed(seed)
    plan, _ = unet_plan(**params)
    state = {}

    def conv(key, cout, cin, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        state[key + '.weight'] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        state[key + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    for op in plan:
        if op['kind'] == 'unit':
            conv(op['key'] + '.conv', op['cout'], op['cin'], 3)
            if op['bn']:
                c = op['cout']
                state[op['key'] + '.bn.weight'] = torch.rand(c, generator=g) * 0.5 + 0.75
                state[op['key'] + '.bn.bias'] = torch.randn(c, generator=g) * 0.1
                state[op['key'] + '.bn.running_mean'] = torch.randn(c, generator=g) * 0.1
                state[op['key'] + '.bn.running_var'] = torch.rand(c, generator=g) + 0.5
                state[op['key'] + '.bn.num_batches_tracked'] = torch.tensor(1)
        elif op['kind'] == 'up':
            conv(op['key'], op['cout'], op['cin'], 3)
        elif op['kind'] in ('head', 'res_add'):
            conv(op['key'], op['cout'], op['cin'], 1)
    return state


def reference_init_state(seed, bn_seed=None, **params):
    """Weights exactly as ``torch.manual_seed(seed); UNet(**params)`` of the reference draws them:
    torch's default nn.Conv2d initialisers consumed in the reference's construction order
    (unet.py:133-164 -- note that an up level builds its block BEFORE its upconv conv because the
    block is a constructor argument, unet.py:155), BatchNorm at its defaults.  With ``bn_seed`` the
    BN affine/running stats are then randomised the way tests/golden/generate_golden.py does for
    its fixtures.  Lets the GPU box rebuild the full-width G11 weights without the reference."""
    import torch.nn as nn
    plan, _ = unet_plan(**params)
    # construction order: trunk as executed, except (block, then upconv) inside every up level
    order = []
    i = 0
    while i < len(plan):
        op = plan[i]
        if op['kind'] == 'up':
            order.extend([plan[i + 1], plan[i + 2], op])
            i += 3
        else:
            if op['kind'] in ('unit', 'head'):
                order.append(op)
            i += 1
    torch.manual_seed(seed)
    state = {}
    convs = {}
    for op in order:
        k = 3 if op['kind'] != 'head' else 1
        convs[op['key']] = nn.Conv2d(op['cin'], op['cout'], k, padding=k // 2)
    gen = torch.Generator().manual_seed(bn_seed) if bn_seed is not None else None
    for op in plan:  # state_dict / module order == execution order
        if op['kind'] not in ('unit', 'up', 'head'):
            continue
        conv = convs[op['key']]
        if op['kind'] == 'unit':
            state[op['key'] + '.conv.weight'] = conv.weight.detach()
            state[op['key'] + '.conv.bias'] = conv.bias.detach()
            if op['bn']:
                c = op['cout']
                mean, var, w, b = torch.zeros(c), torch.ones(c), torch.ones(c), torch.zeros(c)
                if gen is not None:
                    mean = torch.randn(c, generator=gen) * 0.2
                    var = torch.rand(c, generator=gen) + 0.5
                    w = torch.rand(c, generator=gen) + 0.5
                    b = torch.randn(c, generator=gen) * 0.2
                    w[::5] *= -1.0
                state[op['key'] + '.bn.weight'] = w
                state[op['key'] + '.bn.bias'] = b
                state[op['key'] + '.bn.running_mean'] = mean
                state[op['key'] + '.bn.running_var'] = var
                state[op['key'] + '.bn.num_batches_tracked'] = torch.tensor(0)
        else:
            state[op['key'] + '.weight'] = conv.weight.detach()
            state[op['key'] + '.bias'] = conv.bias.detach()
    return state


def stress_state(state, bn_gain, head_gain):
    """The numerics stress weights of golden fixture G18: every BatchNorm affine (weight and bias) times ``bn_gain`` -- activations then
    grow from unit to unit instead of shrinking as they do under torch's default initialisation: with 2.5 the interior activations of
    the full-width U-Net reach 1e2..1e3 -- and the 1x1 classifier (conv_cls.1) times ``head_gain``, which sets the range of the logits
    (+-20 under Dropout2d(0.3) with 0.5).  What trained checkpoints look like to the kernels: wide activation ranges, saturated and
    unsaturated softmax inputs side by side.  tests/golden/generate_golden.py applies the same rule to the reference's own module."""
    out = {}
    for k, v in state.items():
        v = torch.as_tensor(v).clone()
        if k.endswith('.bn.weight') or k.endswith('.bn.bias'):
            v = v * bn_gain
        if k.startswith('conv_cls.1.'):
            v = v * head_gain
        out[k] = v
    return out
