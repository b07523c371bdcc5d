"""Model seam: a drop-in for the reference's ``UNet`` module whose forward runs on librcu_hip.

Mirrors ``common/model/unet.py:123-186`` of the reference at the interface level:
  * same constructor signature and defaults (unet.py:124-130);
  * same ``state_dict`` keys and shapes, so ``load_state_dict(torch.load(ckpt)['state_dict'])``
    (common/model/management.py:56-64) works unchanged -- a ``module.`` prefix is stripped;
  * same ``Dropout2d`` sub-modules under the same names, so ``set_dropout_mode(model, True)``
    (common/utils/torchhelper.py:44-50) switches MC-dropout on exactly as for the reference;
  * ``model(images)`` with ``images`` float32 ``[N, Cin, H, W]`` on the GPU returns ``logits`` or
    ``(logits, sigma)`` (unet.py:181-186).
The sub-modules only hold parameters; all arithmetic happens in the HIP kernels (conv units with
folded eval-mode BatchNorm, dropout as per-(sample, channel) factors drawn here with torch's device
generator like ``feature_dropout`` does).  Inference only: no autograd, BatchNorm always in eval
mode (the test context calls ``model.eval()``, common/trainloop/context.py:321).
"""
import ctypes
import itertools

import torch
import torch.nn as nn

from . import _lib


class _Holder(nn.Module):
    """Plain container; children are added by dotted path to reproduce the reference's key names."""


def _add(root, path, module):
    parts = path.split('.')
    cur = root
    for p in parts[:-1]:
        if p not in cur._modules:
            cur.add_module(p, _Holder())
        cur = cur._modules[p]
    cur.add_module(parts[-1], module)


def _dropout_rule(dropout_center, level, depth, is_down):
    # unet.py:74-82
    if dropout_center is None:
        return 'all'
    if level == depth:
        return 'no'
    if level + dropout_center >= depth:
        return 'last' if is_down else 'first'
    return 'no'


def _has_dropout(dropout, rule, i):
    # unet.py:63-72 (two repetitions per block)
    if dropout is None:
        return False
    return rule == 'all' or (rule == 'first' and i == 0) or (rule == 'last' and i == 1)


class UNet(nn.Module):
    DEFAULT_DEPTH = 4
    DEFAULT_START_FILTERS = 16
    DEFAULT_DROPOUT = 0.2
    MAX_HANDLES = 6          # cached (height, width[, lane]) plans incl. their workspaces
    # rcu_unet_options (include/rcu.h): what the planner may choose.  The defaults are the shipped path; ``plan_options`` of an instance
    # overrides them for A/B measurements and for the parity tests that compare kernel families on the same input.
    PLAN_DEFAULTS = dict(conv_winograd=1, conv_winograd4=1, conv_first=1, act_layout=0, head_winograd4=1, pad_levels=1)
    _generations = itertools.count(1)      # every plan ever created gets the next number: a borrower's plan is valid for ONE generation of its donor's

    def __init__(self, nb_classes, in_channels, depth=DEFAULT_DEPTH, start_filters=DEFAULT_START_FILTERS,
                 dropout=DEFAULT_DROPOUT, dropout_center: int = None, residual=False, sigma_out=False,
                 provide_features=False, bn=True):
        super().__init__()
        self.nb_classes, self.in_channels, self.depth = nb_classes, in_channels, depth
        self.start_filters, self.dropout, self.dropout_center = start_filters, dropout, dropout_center
        self.sigma_out, self.bn, self.residual = sigma_out, bn, bool(residual)
        # unet.py:135-136, 178-179: when set, ``features`` is the input of conv_cls after every forward --
        # here a [N, C, H, W] VIEW of the handle's channels-last workspace tensor (no copy), valid until the
        # next forward (``.clone()`` it to keep it; it is reset to None when the plan it points into is rebuilt or
        # dropped); rcu_amd.model.PostNet consumes it in place
        self.provide_features = provide_features
        self.features = None
        self._site_modules = []
        self._sites = None

        def unit(prefix, cin, cout, with_dropout):
            base = prefix + '.conv2d_batch_relu'
            _add(self, base + '.conv', nn.Conv2d(cin, cout, 3, padding=1))
            if with_dropout:
                do = nn.Dropout2d(p=dropout)
                _add(self, base + '.dropout', do)
                self._site_modules.append(do)
            if bn:
                _add(self, base + '.bn', nn.BatchNorm2d(cout))

        def block(prefix, cin, cout, rule):
            for i in range(2):
                unit('{}.{}'.format(prefix, i), cin if i == 0 else cout, cout, _has_dropout(dropout, rule, i))
            if residual:   # ConvResidualBlock (unet.py:42-60): "<block>.residual", a 1x1 conv of the block input
                _add(self, prefix[:-len('.block')] + '.residual', nn.Conv2d(cin, cout, 1))

        cin, cout = in_channels, start_filters
        for lvl in range(depth):
            block('down_convs.{}.block.block'.format(lvl), cin, cout, _dropout_rule(dropout_center, lvl, depth, True))
            cin, cout = cout, cout * 2
        block('bottom_convs.block', cin, cout, _dropout_rule(dropout_center, depth, depth, True))
        for j, lvl in enumerate(range(depth - 1, -1, -1)):
            cin, cout = cout, cout // 2
            block('up_convs.{}.block.block'.format(j), 2 * cout, cout,
                  _dropout_rule(dropout_center, lvl, depth, False))
            _add(self, 'up_convs.{}.upconv.1'.format(j), nn.Conv2d(cin, cout, 3, padding=1))
        unit('conv_cls.0', cout, cout, dropout is not None)
        _add(self, 'conv_cls.1', nn.Conv2d(cout, nb_classes, 1))
        if sigma_out:
            unit('conv_sigma.0', cout, cout, dropout is not None)
            _add(self, 'conv_sigma.1', nn.Conv2d(cout, nb_classes, 1))
        for p in self.parameters():
            p.requires_grad = False
        self._handles = {}       # (H, W[, lane][, 'features'], plan options) -> (handle, max_batch, weights version, donor handle)
        self._weights_version = 0
        self.plan_options = {}   # overrides of PLAN_DEFAULTS; part of the plan's cache key
        self.fuse_head = True    # 1x1 classifier + softmax + statistics inside conv_cls.0's epilogue where the shapes allow
        self._donor = None       # share_workspace(): the model whose activation workspaces this one's plans borrow
        self._last_generation = 0   # generation of the plan _handle returned last
        self.eval()

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict=True, **kwargs):
        state_dict = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        result = super().load_state_dict(state_dict, strict=strict, **kwargs)
        self.weights_changed()
        return result

    def weights_changed(self):
        """Call after editing parameters in place: the packed device copies are rebuilt lazily."""
        self._weights_version += 1

    def share_workspace(self, donor):
        """Borrow the activation workspaces of ``donor`` -- a UNet of the same architecture -- instead of allocating 6 GB per 160
        slices again: the members of an ensemble (bin-dl/brats_test_ensemble.py:44-57) differ in their 35 MB of packed weights only
        (include/rcu.h: rcu_unet_create_with).  Models that share a workspace must run on ONE stream at a time (the same stream
        lane); ``None`` undoes it.  Existing plans are dropped."""
        if donor is self:
            donor = None
        if donor is not None:
            mine = (self.nb_classes, self.in_channels, self.depth, self.start_filters, self.dropout is not None, self.dropout_center,
                    self.residual, self.sigma_out, bool(self.provide_features), self.bn)
            theirs = (donor.nb_classes, donor.in_channels, donor.depth, donor.start_filters, donor.dropout is not None,
                      donor.dropout_center, donor.residual, donor.sigma_out, bool(donor.provide_features), donor.bn)
            if mine != theirs:
                raise ValueError('share_workspace: the donor is a different architecture')
            if donor._donor is not None:
                donor = donor._donor         # one level: everybody borrows from the owner
            if donor is self:                # (the owner asked to borrow from one of its own borrowers: it stays the owner)
                return
        self._release()
        self._donor = donor

    def set_fuse_head(self, on):
        """Run-time switch between the fused classifier head and the standalone head kernel (same bits; benchmarks time the latter)."""
        self.fuse_head = bool(on)
        lib = _lib.load()
        for entry in self._handles.values():
            _lib.check(lib.rcu_unet_set_fuse_head(entry[0], int(self.fuse_head)))

    def _options(self):
        unknown = set(self.plan_options) - set(self.PLAN_DEFAULTS)
        if unknown:
            raise ValueError('unknown plan options: {}'.format(sorted(unknown)))
        return dict(self.PLAN_DEFAULTS, **self.plan_options)

    def _release(self):
        lib = _lib.load()
        for entry in self._handles.values():
            lib.rcu_unet_destroy(entry[0])
        self._handles = {}
        self.features = None     # a view into a destroyed handle's workspace

    def __del__(self):
        try:
            self._release()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _handle(self, h, w, n, lane=0):
        """The library handle (plan + packed weights + activation workspace) for images of h x w, batches up to n.  ``lane``: launches
        that run concurrently on different HIP streams (rcu_amd.distributed: stream lanes) need a workspace each; a lane is a
        second, third ... handle of the same shape."""
        lib = _lib.load()
        options = self._options()
        slot = (h, w) if lane == 0 else (h, w, lane)
        if self.provide_features:      # a plan of its own: the feature tensor is kept channels-last (include/rcu.h, rcu_unet_desc)
            slot = slot + ('features',)
        slot = slot + (tuple(sorted(options.items())),)
        entry = self._handles.get(slot)
        donor_handle = None
        donor_key = None
        if self._donor is not None:    # the donor's plan of this shape and lane first (it may have to grow): its workspace is ours
            donor_handle = self._donor._handle(h, w, n if entry is None else max(n, entry[1]), lane)
            # (the plan's generation, not the handle's address: a re-created plan often gets the freed one's address, and a borrower that
            # kept its old plan would keep the OLD workspace alive and share nothing)
            donor_key = self._donor._last_generation
        if entry is not None and entry[1] >= n and entry[2] == self._weights_version and entry[3] == donor_key:
            self._handles[slot] = self._handles.pop(slot)     # most recently used last
            self._last_generation = entry[4]
            return entry[0]
        max_batch = n if entry is None else max(n, entry[1])
        if entry is not None:
            lib.rcu_unet_destroy(entry[0])
            del self._handles[slot]
            self.features = None     # ``features`` may be a view into that handle's workspace: valid until the next forward only
        desc = _lib.UnetDesc(nb_classes=self.nb_classes, in_channels=self.in_channels, depth=self.depth,
                             start_filters=self.start_filters, has_dropout=int(self.dropout is not None),
                             dropout_center=-1 if self.dropout_center is None else int(self.dropout_center),
                             sigma_out=int(self.sigma_out), bn=int(self.bn), height=h, width=w, max_batch=max_batch,
                             residual=int(self.residual), provide_features=int(bool(self.provide_features)))
        opts = _lib.UnetOptions(fuse_head=int(self.fuse_head), **options)
        handle = ctypes.c_void_p()
        _lib.check(lib.rcu_unet_create_with(ctypes.byref(desc), ctypes.byref(opts), donor_handle, ctypes.byref(handle)))
        try:
            for key, value in self.state_dict().items():
                if not torch.is_floating_point(value):
                    continue  # num_batches_tracked
                host = value.detach().to('cpu', torch.float32).contiguous()
                _lib.check(lib.rcu_unet_load_weight(handle, key.encode(), ctypes.c_void_p(host.data_ptr()),
                                                    host.numel()))
            _lib.check(lib.rcu_unet_finalize_weights(handle))
        except Exception:
            lib.rcu_unet_destroy(handle)
            raise
        self._last_generation = next(UNet._generations)
        self._handles[slot] = (handle, max_batch, self._weights_version, donor_key, self._last_generation)
        while len(self._handles) > self.MAX_HANDLES:    # images of many different sizes: drop the least recently used plan
            old = next(iter(self._handles))
            lib.rcu_unet_destroy(self._handles.pop(old)[0])
            self.features = None
        return handle

    def max_group_samples(self, h, w):
        """Largest batch of h x w images whose every activation tensor stays below the 2 GB that the Winograd kernels' 32-bit buffer offsets
        reach (csrc/rcu_api.hip pick_config: beyond it a layer falls back to the direct kernels -- correct, but slower).  The widest
        full-resolution tensor has ``start_filters`` channels (twice that for the classifier + sigma twin unit)."""
        # (the planner pads every tensor's channels to a multiple of 32 -- pick_config's bound is on the PADDED tensor)
        widest = max(8, (self.start_filters + 31) // 32 * 32 * (2 if self.sigma_out else 1))
        # ... and, where the image is not a whole number of 32 x 32 tiles, level 0 may be ALLOCATED with an extent rounded up to them
        # (rcu_unet_options.pad_levels; the reference's 240 x 240 BraTS slices: 256 x 256 at most) -- the bound is on that tensor
        h, w = int(h), int(w)
        if self._options().get('pad_levels', 1) and (h % 32 or w % 32):
            h, w = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        return max(1, ((1 << 31) - 1) // (h * w * 4 * widest))

    def reserve(self, h, w, n, lane=0):
        """Make the plan and the activation workspace for batches of up to ``n`` images of h x w now (a step that knows it will run pass
        groups of n * g samples calls this before its first, smaller forward: one plan instead of a small one that is replaced)."""
        self._handle(h, w, n, lane)

    # ------------------------------------------------------------------ dropout
    def dropout_sites(self):
        """[(state_dict-style name, channels)] in execution order."""
        if self._sites is None:        # the module tree is fixed after construction: walk it once
            modules = dict(self.named_modules())
            names = {id(m): n for n, m in modules.items()}
            self._sites = [(names[id(m)], modules[names[id(m)][:-len('.dropout')] + '.conv'].out_channels)
                           for m in self._site_modules]
        return list(self._sites)

    def mc_active(self):
        return any(m.training for m in self._site_modules)

    def sample_masks(self, n, device, generator=None):
        """Concatenated ``[site][n][C_site]`` factors {0, 1/(1-p)} for one pass; sites whose Dropout2d
        is in eval mode get ones.  Same law as torch's feature dropout (Bernoulli(1-p) / (1-p))."""
        # consecutive sites with the same state share one Bernoulli draw (the elements are i.i.d.): for the shipped configs
        # -- every site active with one p -- a pass costs two small kernels instead of three per site
        groups = []   # [p or None (inactive), number of factors]
        for m, (_, c) in zip(self._site_modules, self.dropout_sites()):
            state = float(m.p) if (m.training and m.p > 0) else None
            if groups and groups[-1][0] == state:
                groups[-1][1] += n * c
            else:
                groups.append([state, n * c])
        chunks = []
        for state, count in groups:
            if state is None:
                chunks.append(torch.ones(count, device=device))
            elif state >= 1:
                chunks.append(torch.zeros(count, device=device))
            else:
                keep = 1.0 - state
                chunks.append(torch.empty(count, device=device).bernoulli_(keep, generator=generator).div_(keep))
        if not chunks:
            return None
        return chunks[0] if len(chunks) == 1 else torch.cat(chunks)

    def seeded_masks(self, n, device, seeds, first_sample=0):
        """The Dropout2d factors of ``len(seeds)`` MC passes over n images, drawn on the device by ONE kernel (include/rcu.h,
        rcu_dropout_masks) on the current stream: the factors of image i in pass t from ``seeds[t]`` and the image's GLOBAL index
        ``first_sample + i`` alone -- the same values whatever batch the image arrives in and whatever group, lane or rank the pass is
        launched in -- in the ``[site][passes * n][C_site]`` layout ``forward_accumulate(..., passes=len(seeds))`` reads (one pass: the layout
        of ``sample_masks``).  Sites whose Dropout2d is in eval mode get ones, p = 1 zeros, as ``sample_masks`` gives them."""
        sites = self.dropout_sites()
        if not sites:
            return None
        g, count = len(seeds), len(sites)
        keeps = [(max(0.0, 1.0 - float(m.p)) if (m.training and m.p > 0) else -1.0) for m in self._site_modules]
        out = torch.empty(g * n * sum(c for _, c in sites), device=device, dtype=torch.float32)
        _lib.check(_lib.load().rcu_dropout_masks((ctypes.c_uint64 * g)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds]), g, n, int(first_sample),
                                                 (ctypes.c_int32 * count)(*[c for _, c in sites]), (ctypes.c_float * count)(*keeps), count,
                                                 _lib.ptr(out), _lib.current_stream()))
        return out

    def pack_masks(self, masks, n, device):
        """List of per-site ``[n, C_site]`` arrays/tensors -> the concatenated device layout."""
        sites = self.dropout_sites()
        if len(masks) != len(sites):
            raise ValueError('expected {} dropout masks, got {}'.format(len(sites), len(masks)))
        flat = []
        for m, (_, c) in zip(masks, sites):
            t = torch.as_tensor(m, dtype=torch.float32)
            if tuple(t.shape) != (n, c):
                raise ValueError('mask shape {} does not match (n={}, channels={})'.format(tuple(t.shape), n, c))
            flat.append(t.reshape(-1))
        return torch.cat(flat).to(device).contiguous() if flat else None

    # ------------------------------------------------------------------ forward
    def _check_input(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError('expected a [N, {}, H, W] tensor'.format(self.in_channels))
        if not x.is_cuda:
            raise RuntimeError('rcu_amd.model.UNet only runs on the GPU (librcu_hip); got a {} tensor'.format(x.device))
        return x.to(torch.float32).contiguous()

    def forward(self, x, masks=None):
        """``masks``: None -> eval mode, or sampled if any Dropout2d is in train mode
        (set_dropout_mode); a concatenated device tensor / list of per-site arrays to inject."""
        x = self._check_input(x)
        n, _, h, w = x.shape
        handle = self._handle(h, w, n)
        if masks is None and self.mc_active():
            masks = self.sample_masks(n, x.device)
        elif isinstance(masks, (list, tuple)):
            masks = self.pack_masks(masks, n, x.device)
        logits = torch.empty((n, self.nb_classes, h, w), device=x.device, dtype=torch.float32)
        sigma = torch.empty_like(logits) if self.sigma_out else None
        _lib.check(_lib.load().rcu_unet_forward(handle, _lib.ptr(x), n, _lib.ptr(masks), _lib.ptr(logits),
                                                _lib.ptr(sigma), _lib.current_stream()))
        if self.provide_features:
            self.features = self._features_view(handle, n, h, w, x.device)
        if self.sigma_out:
            return logits, sigma
        return logits

    def _features_view(self, handle, n, h, w, device):
        ptr, ch, pitch = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().rcu_unet_features(handle, ctypes.byref(ptr), ctypes.byref(ch), ctypes.byref(pitch)))
        nhwc = _lib.device_view(ptr.value, (n, h, w, pitch.value), device, owner=self)
        return nhwc[..., :ch.value].permute(0, 3, 1, 2)

    def forward_accumulate(self, x, stats, masks=None, passes=1, lane=0):
        """One pass -- or ``passes`` stochastic passes as ONE batch of N * passes samples -- fused with softmax +
        accumulation into ``stats`` (rcu_amd.steps.McStatistics): neither logits nor probabilities reach HBM.
        ``masks`` for a pass group: a concatenated device tensor with N * passes rows per site, or a list of
        ``passes`` mask sets (each a concatenated tensor or a list of per-site ``[N, C_site]`` arrays)."""
        x = self._check_input(x)
        n, _, h, w = x.shape
        if (n, self.nb_classes, h * w) != (stats.n, stats.nb_classes, stats.hw):
            raise ValueError('statistics blob shape does not match the batch')
        if passes < 1:
            raise ValueError('passes must be >= 1')
        handle = self._handle(h, w, n * passes, lane)
        if passes == 1:
            if masks is None and self.mc_active():
                masks = self.sample_masks(n, x.device)
            elif isinstance(masks, (list, tuple)):
                masks = self.pack_masks(masks, n, x.device)
            _lib.check(_lib.load().rcu_unet_forward_accumulate(handle, _lib.ptr(x), n, _lib.ptr(masks),
                                                               _lib.ptr(stats.blob), stats.flags, _lib.current_stream()))
        else:
            if masks is None:
                if not self.mc_active():
                    raise ValueError('a pass group needs stochastic passes: set_dropout_mode(model, True) or inject masks')
                masks = self.sample_masks(n * passes, x.device)
            elif isinstance(masks, (list, tuple)):
                masks = self.group_masks(masks, n, x.device)
            _lib.check(_lib.load().rcu_unet_forward_accumulate_passes(handle, _lib.ptr(x), n, passes, _lib.ptr(masks),
                                                                      _lib.ptr(stats.blob), stats.flags,
                                                                      _lib.current_stream()))
        stats.count += passes

    def forward_accumulate_sigma(self, x, stats, sigma_sum, masks=None, is_log_sigma=False, lane=0, passes=1):
        """EXTENSION (BASELINE config "aleatoric + MC", not in the reference): one stochastic pass of a ``sigma_out`` model --
        softmax(logits) into ``stats`` as in ``forward_accumulate`` and this pass's sigma (|raw|, or exp(raw)) added to the
        float32 ``[N, C, H, W]`` tensor ``sigma_sum``; neither logits nor sigma reach HBM as volumes of their own."""
        if not self.sigma_out:
            raise ValueError('forward_accumulate_sigma needs a model built with sigma_out=True')
        x = self._check_input(x)
        n, _, h, w = x.shape
        if (n, self.nb_classes, h * w) != (stats.n, stats.nb_classes, stats.hw):
            raise ValueError('statistics blob shape does not match the batch')
        if (tuple(sigma_sum.shape) != (n, self.nb_classes, h, w) or sigma_sum.dtype != torch.float32 or
                not sigma_sum.is_contiguous() or sigma_sum.device != x.device):
            raise ValueError('sigma_sum must be a contiguous float32 [N, C, H, W] tensor on the input device')
        if passes < 1:
            raise ValueError('passes must be >= 1')
        handle = self._handle(h, w, n * passes, lane)
        if passes > 1:      # pass group (see forward_accumulate): ``masks`` = a list of ``passes`` mask sets or a concatenated tensor
            if masks is None:
                if not self.mc_active():
                    raise ValueError('a pass group needs stochastic passes: set_dropout_mode(model, True) or inject masks')
                masks = self.sample_masks(n * passes, x.device)
            elif isinstance(masks, (list, tuple)):
                masks = self.group_masks(masks, n, x.device)
            _lib.check(_lib.load().rcu_unet_forward_accumulate_sigma_passes(handle, _lib.ptr(x), n, passes, _lib.ptr(masks),
                                                                            _lib.ptr(stats.blob), stats.flags, _lib.ptr(sigma_sum),
                                                                            int(bool(is_log_sigma)), _lib.current_stream()))
            stats.count += passes
            return
        if masks is None and self.mc_active():
            masks = self.sample_masks(n, x.device)
        elif isinstance(masks, (list, tuple)):
            masks = self.pack_masks(masks, n, x.device)
        _lib.check(_lib.load().rcu_unet_forward_accumulate_sigma(handle, _lib.ptr(x), n, _lib.ptr(masks), _lib.ptr(stats.blob),
                                                                 stats.flags, _lib.ptr(sigma_sum), int(bool(is_log_sigma)),
                                                                 _lib.current_stream()))
        stats.count += 1

    def group_masks(self, mask_sets, n, device):
        """``passes`` mask sets (each [site][n][C_site]) -> the [site][passes * n][C_site] layout of a pass group."""
        sites = self.dropout_sites()
        per_set = []
        for ms in mask_sets:
            flat = self.pack_masks(ms, n, device) if isinstance(ms, (list, tuple)) else ms.to(device)
            per_set.append(torch.split(flat, [n * c for _, c in sites]) if sites else [])
        chunks = [torch.cat([ps[i] for ps in per_set]) for i in range(len(sites))]
        return torch.cat(chunks).contiguous() if chunks else None

    # ------------------------------------------------------------------ introspection (bench)
    def layer_table(self, h, w, n=1):
        lib = _lib.load()
        handle = self._handle(h, w, n)
        rows = []
        for i in range(lib.rcu_unet_num_layers(handle)):
            info = _lib.LayerInfo()
            _lib.check(lib.rcu_unet_layer_info(handle, i, ctypes.byref(info)))
            rows.append(dict(index=i, name=info.name.decode(), kernel=info.kernel.decode(), cin=info.cin,
                             cout=info.cout, height=info.height, width=info.width, grid_height=info.grid_height, grid_width=info.grid_width,
                             upsample=bool(info.upsample),
                             pooled=bool(info.pooled), dual_source=bool(info.dual_source), head_fusable=bool(info.head_fusable),
                             flops_per_slice=info.flops_per_slice,
                             mfma_flops_per_slice=info.mfma_flops_per_slice))
        return rows

    def run_layer(self, h, w, n, layer, masks=None):
        _lib.check(_lib.load().rcu_unet_run_layer(self._handle(h, w, n), layer, n, _lib.ptr(masks),
                                                  _lib.current_stream()))

    def profile_begin(self, h, w, n, max_forwards):
        """Time every kernel of the next ``max_forwards`` forwards with HIP events on the current stream."""
        _lib.check(_lib.load().rcu_unet_profile_begin(self._handle(h, w, n), max_forwards))

    def profile_collect(self, h, w, n):
        """-> (forwards covered, [summed ms per slot]): slot 0 input re-layout, 1..L conv layers, L+1 head."""
        lib = _lib.load()
        handle = self._handle(h, w, n)
        slots = lib.rcu_unet_num_layers(handle) + 2
        ms = (ctypes.c_double * slots)()
        cnt = ctypes.c_int()
        _lib.check(lib.rcu_unet_profile_collect(handle, ms, ctypes.byref(cnt)))
        return cnt.value, list(ms)

    def workspace_bytes(self, h, w, n, lane=0):
        """Device bytes the plan OWNS: packed weights, plus the activation workspace unless it is borrowed (share_workspace)."""
        return int(_lib.load().rcu_unet_workspace_bytes(self._handle(h, w, n, lane)))


class PostNet(nn.Module):
    """Drop-in for common/model/postnet.py:6-18 (the auxiliary confidence network of the auxiliary_feat runs):
    ``nb_convs`` x [Conv2d 1x1 + BatchNorm2d + ReLU] + Conv2d 1x1 -> nb_classes, same constructor and state_dict
    keys, evaluated by ONE fused kernel (csrc/rcu_postnet.hip).  Fed with ``UNet.features`` it reads the U-Net's
    channels-last workspace tensor in place; any other [N, C, H, W] device tensor is re-laid out first."""

    def __init__(self, in_channels, nb_classes, nb_convs=3, dropout=None):
        super().__init__()
        self.in_channels, self.nb_classes, self.nb_convs = in_channels, nb_classes, nb_convs
        self._dropouts = []
        for i in range(nb_convs):
            base = 'convs.{}.conv2d_batch_relu'.format(i)
            _add(self, base + '.conv', nn.Conv2d(in_channels, in_channels, 1))
            if dropout is not None:
                do = nn.Dropout2d(p=dropout)
                _add(self, base + '.dropout', do)
                self._dropouts.append(do)
            _add(self, base + '.bn', nn.BatchNorm2d(in_channels))
        _add(self, 'conv_logits', nn.Conv2d(in_channels, nb_classes, 1))
        for p in self.parameters():
            p.requires_grad = False
        self._handle_entry = None    # (handle, weights version)
        self._weights_version = 0
        self.eval()

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        state_dict = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        result = super().load_state_dict(state_dict, strict=strict, **kwargs)
        self.weights_changed()
        return result

    def weights_changed(self):
        self._weights_version += 1

    def __del__(self):
        try:
            if self._handle_entry is not None:
                _lib.load().rcu_postnet_destroy(self._handle_entry[0])
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _handle(self):
        lib = _lib.load()
        if self._handle_entry is not None and self._handle_entry[1] == self._weights_version:
            return self._handle_entry[0]
        if self._handle_entry is not None:
            lib.rcu_postnet_destroy(self._handle_entry[0])
            self._handle_entry = None
        handle = ctypes.c_void_p()
        _lib.check(lib.rcu_postnet_create(self.in_channels, self.nb_classes, self.nb_convs, 1, ctypes.byref(handle)))
        try:
            for key, value in self.state_dict().items():
                if not torch.is_floating_point(value):
                    continue
                host = value.detach().to('cpu', torch.float32).contiguous()
                _lib.check(lib.rcu_postnet_load_weight(handle, key.encode(), ctypes.c_void_p(host.data_ptr()),
                                                       host.numel()))
            _lib.check(lib.rcu_postnet_finalize_weights(handle))
        except Exception:
            lib.rcu_postnet_destroy(handle)
            raise
        self._handle_entry = (handle, self._weights_version)
        return handle

    def sample_masks(self, n, device, generator=None):
        """Dropout2d factors {0, 1/(1-p)} of one pass, ``[nb_convs][n][C]`` (ones where a module is in eval mode); None when the
        network has no dropout modules or none of them is in train mode."""
        if not self._dropouts or not any(m.training for m in self._dropouts):
            return None
        rows = []
        for m in self._dropouts:
            if m.training and 0 < m.p < 1:
                keep = 1.0 - float(m.p)
                rows.append(torch.empty(n * self.in_channels, device=device).bernoulli_(keep, generator=generator).div_(keep))
            elif m.training and m.p >= 1:
                rows.append(torch.zeros(n * self.in_channels, device=device))
            else:
                rows.append(torch.ones(n * self.in_channels, device=device))
        return torch.cat(rows)

    def forward(self, x, masks=None):
        """``masks``: None -> eval mode, or sampled if any Dropout2d is in train mode (set_dropout_mode); a list of per-conv
        ``[N, C]`` arrays / a concatenated device tensor to inject."""
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError('expected a [N, {}, H, W] tensor'.format(self.in_channels))
        if not x.is_cuda:
            raise RuntimeError('rcu_amd.model.PostNet only runs on the GPU (librcu_hip); got a {} tensor'.format(x.device))
        n, c, h, w = x.shape
        if masks is None:
            masks = self.sample_masks(n, x.device)
        elif isinstance(masks, (list, tuple)):
            if len(masks) != self.nb_convs:
                raise ValueError('expected {} dropout masks, got {}'.format(self.nb_convs, len(masks)))
            masks = torch.cat([torch.as_tensor(m, dtype=torch.float32).reshape(n * c) for m in masks]).to(x.device)
        if masks is not None:
            masks = masks.to(torch.float32).contiguous()
            if masks.numel() != self.nb_convs * n * c:
                raise ValueError('dropout masks must hold nb_convs x N x C factors')
        cp = (c + 31) // 32 * 32
        nhwc = x.permute(0, 2, 3, 1)
        pitch = nhwc.stride(2)
        in_place = (x.dtype == torch.float32 and nhwc.stride(3) == 1 and pitch >= cp and pitch % 4 == 0 and
                    nhwc.stride(1) == w * pitch and nhwc.stride(0) == h * w * pitch and x.data_ptr() % 16 == 0)
        if not in_place:      # generic input: channels-last copy, zero padded to the kernel's voxel pitch (a multiple of 32 floats)
            pitch = cp
            buf = torch.zeros((n, h, w, pitch), device=x.device, dtype=torch.float32)
            buf[..., :c] = nhwc
            nhwc = buf
        logits = torch.empty((n, self.nb_classes, h, w), device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().rcu_postnet_forward(self._handle(), ctypes.c_void_p(nhwc.data_ptr()), pitch, n, h * w,
                                                   _lib.ptr(masks), _lib.ptr(logits), _lib.current_stream()))
        return logits


model_registry = {'unet': UNet, 'postnet': PostNet}  # common/model/factory.py:12-15


def get_model(model_type, **params):
    return model_registry[model_type](**params)
