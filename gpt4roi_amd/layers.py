"""MI355X drop-in for `gpt4roi.models.layers` (the region-feature module).

Mirrors /root/reference/gpt4roi/models/layers.py: `MLVLFuseModule` (:96-195),
`MLVLROIQueryModule` (:198-236) and `MlvlRoIExtractor` (:239-335) keep their constructor
arguments, attribute names and therefore their `state_dict` keys
(`mlvl_fuse.input_conv.N.*`, `mlvl_fuse.fuse_convs.N.conv|gn.*`, `roi_align.pconvs.N.*`,
`roi_align.pos_embedd.{0,2,3,5}.*`, `roi_align.updims.*`, `roi_align.flatten_linear.*`), and
`MLVLROIQueryModule.forward(mlvl_feats, bboxes) -> list[B] of [n_i, out_dims]` keeps its
contract.  The arithmetic runs on the hand-written gfx950 kernels (gpt4roi_amd/csrc) in a
different decomposition than the reference's op-by-op PyTorch graph:

  reference (per level, per round)              here (NHWC, bf16 storage, fp32 math)
  --------------------------------------------  ---------------------------------------------
  interpolate -> cat(coords) -> conv1x1         upsample_coord kernel -> one MFMA GEMM (K padded)
  slice/interp(fp32)/cat -> conv3x3 -> GN -> ReLU  fuse_shuffle gather (applies the PREVIOUS round's
                                                GN+ReLU on the fly) -> implicit-GEMM conv ->
                                                GN statistics only (normalised map never stored)
  4 x roi_align(fp32) on normalised maps        one multi-level NHWC RoIAlign launch that applies
                                                the last GN+ReLU per texel
  4 x conv3x3 + sum + ReLU                      one implicit GEMM with K = 4*9*C (+bias sum, ReLU)
  flatten (c-major) -> Linear                   split-K GEMM on a weight permuted once to NHWC order

Differences from the reference, all parameterised and the reference's values the default:
  * the hard-wired 16x16 / 224 constants (layers.py:220-222, 289-291, 297) are derived from the
    input (P = sqrt(tokens), image side 14*P);
  * `out_dims` is honoured (the reference hard-codes Linear(1024, 4096) and ignores the argument);
  * parameters are converted into kernel-ready bf16 buffers (`prepare()`), re-derived automatically when a parameter
    was written since (version stamps).  Training: `forward_train()` / `backward()` are the hand-written backward of
    every stage (gradients under the reference's state_dict keys and layouts; gpt4roi_amd/train.py drives them), and
    with grad enabled `forward()` wraps exactly that pair in an autograd node, so `out.backward()` fills `.grad` the
    way autograd does for the reference.
There is no CPU path: on a machine without the HIP library or a GPU the module raises.
"""
import math

import torch
import torch.nn as nn

from . import kernels as K
from .roi_align import RoIAlign


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """Parameter container with mmcv ConvModule's layout for (3x3 conv, GN): `conv` without
    bias (conv_module.py:104-105), `gn` = GroupNorm(num_groups, eps 1e-5) (norm.py:101-107)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None,
                 norm_cfg=None):
        super().__init__()
        assert conv_cfg is None and norm_cfg is not None and norm_cfg['type'] == 'GN'
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
        self.gn = nn.GroupNorm(norm_cfg['num_groups'], out_channels, eps=1e-5)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')


class MLVLFuseModule(nn.Module):

    def __init__(self, input_dims=1024, embed_dims=1024, num_levels=3, num_fuse=4):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_fuse = num_fuse
        self.input_dims = input_dims
        self.shuffle_channles = embed_dims // 4
        self.fuse_lvl_list = []
        for lvl in range(num_levels):
            self.fuse_lvl_list.append((lvl, min(lvl + 1, num_levels - 1), max(lvl - 1, 0)))
        self.remain_chs = self.embed_dims - self.shuffle_channles * 2
        self.input_conv = nn.ModuleList([nn.Conv2d(self.input_dims + 2, self.embed_dims, 1)
                                         for _ in range(self.num_levels)])
        self.fuse_convs = nn.ModuleList([
            ConvModule(self.embed_dims, self.embed_dims, 3, stride=1, padding=1, conv_cfg=None,
                       norm_cfg=dict(type='GN', num_groups=64, requires_grad=True))
            for _ in range(self.num_fuse)])
        self._ready = None

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                normal_init(m, std=0.01)

    def prepare(self):
        dev = self.input_conv[0].weight.device
        cin = self.input_dims + 2
        self.cpad = -(-cin // 64) * 64
        bf = getattr(self, "compute_dtype", torch.bfloat16)      # set by MLVLROIQueryModule.set_compute_dtype
        w_in, b_in = [], []
        with torch.no_grad():
            for conv in self.input_conv:
                w = torch.zeros(self.embed_dims, self.cpad, dtype=bf, device=dev)
                w[:, :cin] = conv.weight.reshape(self.embed_dims, cin).to(bf)
                w_in.append(w)
                b_in.append(conv.bias.to(bf).float().contiguous())
            w_f = [K.prep_conv3x3_weight(m.conv.weight.detach(), bf) for m in self.fuse_convs]
            gn = [(m.gn.weight.detach().float().contiguous(), m.gn.bias.detach().float().contiguous(),
                   m.gn.num_groups, m.gn.eps) for m in self.fuse_convs]
        self._ready = dict(w_in=w_in, b_in=b_in, w_f=w_f, gn=gn)

    def forward(self, tokens, P, sizes):
        """tokens: list[num_levels] of [B, P*P, C] bf16 (row/batch strided views allowed).
        Returns (raw conv maps [B,H_l,W_l,C] bf16, deferred GN+ReLU affines [B,2,C] fp32)."""
        if self._ready is None:
            self.prepare()
        r = self._ready
        B = tokens[0].size(0)
        maps, affs = [], [None] * self.num_levels
        for lvl, tok in enumerate(tokens):
            H = sizes[lvl]
            x = K.upsample_coord(tok, P, P, H, H, self.cpad)
            y = K.gemm(x.view(B * H * H, self.cpad), r['w_in'][lvl], bias=r['b_in'][lvl])
            maps.append(y.view(B, H, H, self.embed_dims))
        # One implicit-GEMM launch per round over ALL levels (the reference applies the same ConvModule to every level):
        # the levels' maps live stacked in one buffer, 192 x 4 = 768 tiles = three full waves of the 256 CUs for the 336^2
        # pyramid, instead of four launches that each leave a partly idle tail (1219 -> ~740 us per round).
        hw = [(m.size(1), m.size(2)) for m in maps]
        dev = maps[0].device
        for rnd in range(self.num_fuse):
            g, bt, groups, eps = r['gn'][rnd]
            inp = K.MlvlMaps(B, hw, self.embed_dims, dev, dtype=maps[0].dtype)
            K.fuse_shuffle_mlvl(maps, affs, self.fuse_lvl_list, inp)       # every level's conv input: one launch
            z = K.conv3x3_mlvl(inp, r['w_f'][rnd])
            maps = z.levels
            affs = K.groupnorm_affine_mlvl(z, g, bt, groups, eps)          # every level's deferred GN: two launches
        return maps, affs


    # ---- training rows ----------------------------------------------------------------
    def forward_train(self, tokens, P, sizes):
        """Same launches as forward(); additionally keeps the 1x1-conv inputs and every round's raw maps and
        folded GN affines for backward()."""
        if self._ready is None:
            self.prepare()
        r = self._ready
        B = tokens[0].size(0)
        xs, maps = [], []
        for lvl, tok in enumerate(tokens):
            H = sizes[lvl]
            x = K.upsample_coord(tok, P, P, H, H, self.cpad)
            y = K.gemm(x.view(B * H * H, self.cpad), r['w_in'][lvl], bias=r['b_in'][lvl])
            xs.append(x)
            maps.append(y.view(B, H, H, self.embed_dims))
        all_maps, all_affs, inps = [maps], [[None] * self.num_levels], []
        hw = [(m.size(1), m.size(2)) for m in maps]
        dev = maps[0].device
        for rnd in range(self.num_fuse):
            g, bt, groups, eps = r['gn'][rnd]
            inp = K.MlvlMaps(B, hw, self.embed_dims, dev)          # all levels in one buffer: one conv launch per round
            K.fuse_shuffle_mlvl(all_maps[-1], all_affs[-1], self.fuse_lvl_list, inp)
            z = K.conv3x3_mlvl(inp, r['w_f'][rnd])
            all_maps.append(z.levels)
            all_affs.append(K.groupnorm_affine_mlvl(z, g, bt, groups, eps))
            inps.append(inp.levels)            # the convs' inputs stay resident for their weight gradients (0.8 GB per round
            #                                    at 8 x 336^2: re-deriving them cost 20 shuffle launches, 4.5 ms per step)
        return all_maps[-1], all_affs[-1], dict(xs=xs, maps=all_maps, affs=all_affs, inps=inps, B=B)

    def backward(self, ctx, d_y, on_grad=None):
        """d_y: list[num_levels] of fp32 NHWC gradients w.r.t. the POST GN+ReLU maps of the last round.
        Returns {state_dict key: fp32 gradient in the reference layout}.  (No input gradient: the ViT is frozen.)
        `on_grad(key, grad)` fires as each round's gradients are complete (last round first)."""
        r = self._ready
        C, B = self.embed_dims, ctx['B']
        dev = d_y[0].device
        grads = {}
        hw = [(m.size(1), m.size(2)) for m in ctx['maps'][0]]
        merged = C % 256 == 0                       # all levels of a round in ONE weight-gradient launch (they share the weight)
        if getattr(self, '_wgrad_key', None) != (B, tuple(hw), merged):
            self._wgrad_key = (B, tuple(hw), merged)
            self._wgrad_mlvl = K.ConvWgradNHWC(B, hw, C, C, dev) if merged else None
            self._wgrad_plans = None if merged else [K.ConvWgradPlan(B, h, w, C, C, dev) for h, w in hw]
        for rnd in range(self.num_fuse - 1, -1, -1):
            g, bt, groups, eps = r['gn'][rnd]
            z_r, aff_r = ctx['maps'][rnd + 1], ctx['affs'][rnd + 1]
            prev, paff = ctx['maps'][rnd], ctx['affs'][rnd]
            dgamma = torch.zeros(C, dtype=torch.float32, device=dev)
            dbeta = torch.zeros(C, dtype=torch.float32, device=dev)
            wt = K.conv3x3_dgrad_weight(self.fuse_convs[rnd].conv.weight.detach())
            dW = None
            dzs = K.MlvlMaps(B, [(m.size(1), m.size(2)) for m in z_r], C, dev)
            for tar, top, dow in self.fuse_lvl_list:
                stats = K.groupnorm_stats(z_r[tar], groups, eps)
                dz = K.gn_relu_bwd(z_r[tar], d_y[tar], aff_r[tar], g, stats, dgamma, dbeta, groups, out=dzs.levels[tar])
                if not merged:
                    dW = self._wgrad_plans[tar].wgrad(ctx['inps'][rnd][tar], dz, accumulate_into=dW)   # the levels share the weight
            if merged:
                dW = self._wgrad_mlvl.wgrad(ctx['inps'][rnd], dzs.levels)
            dinps = K.conv3x3_mlvl(dzs, wt).levels          # the input gradients of every level: one implicit GEMM
            grads[f'fuse_convs.{rnd}.conv.weight'] = dW
            grads[f'fuse_convs.{rnd}.gn.weight'] = dgamma
            grads[f'fuse_convs.{rnd}.gn.bias'] = dbeta
            if on_grad is not None:
                for k in (f'fuse_convs.{rnd}.gn.bias', f'fuse_convs.{rnd}.gn.weight', f'fuse_convs.{rnd}.conv.weight'):
                    on_grad(k, grads[k])
            # transpose of the channel shuffle + resampling, gathered per source level (fuse_lvl_list is
            # (l, min(l+1, L-1), max(l-1, 0)) as in layers.py:108-112)
            d_y = [K.fuse_shuffle_bwd_gather(l, dinps) for l in range(self.num_levels)]
        cin = self.input_dims + 2
        for lvl in range(self.num_levels):
            dm = K.cast_bf16(d_y[lvl].view(-1, C))
            x = ctx['xs'][lvl]
            dw = K.linear_wgrad(dm, x.view(-1, self.cpad))
            grads[f'input_conv.{lvl}.weight'] = dw[:, :cin].reshape(C, cin, 1, 1).contiguous()
            grads[f'input_conv.{lvl}.bias'] = K.colsum(dm)
        if on_grad is not None:
            for lvl in range(self.num_levels - 1, -1, -1):
                on_grad(f'input_conv.{lvl}.bias', grads[f'input_conv.{lvl}.bias'])
                on_grad(f'input_conv.{lvl}.weight', grads[f'input_conv.{lvl}.weight'])
        return grads


class PreparedBoxes:
    """Host-side preparation of a request's boxes, done ONCE outside the launch sequence so that the
    per-image kernel sequence has no host<->device traffic and can be captured in a hipGraph:
    the [img_id, box * image_size] RoI table of layers.py:295-302, the raw normalised boxes for
    pos_embedd (layers.py:284-285), the per-image counts and their prefix sums (for the splice)."""

    def __init__(self, bboxes, image_size, device, dtype=torch.bfloat16):
        """dtype: the 16-bit storage type of the request.  torch.float16 reproduces the serving path of the reference, where
        the boxes arrive as `.half()` (app.py:271): `single_img_roi * 224` (layers.py:297) is then a HALF product, rounded
        to fp16 before RoIAlign's `.to(float32)` (layers.py:311).  With bf16 (the training dtype, where the dataset hands
        fp32 boxes) the product stays fp32."""
        self.dtype = dtype
        self.counts = [int(b.size(0)) for b in bboxes]
        self.num_imgs = len(bboxes)
        n = sum(self.counts)
        src = bboxes[0].device if self.num_imgs else torch.device("cpu")
        boxes = torch.cat([b.detach().float() for b in bboxes], 0) if n else torch.zeros(0, 4, device=src)
        img_id = torch.cat([torch.full((c,), float(i), device=src) for i, c in enumerate(self.counts)]) \
            if self.num_imgs else torch.zeros(0, device=src)
        scaled = (boxes.to(torch.float16) * float(image_size)).float() if dtype is torch.float16 else boxes * float(image_size)
        self.rois5 = torch.cat([img_id[:, None], scaled], 1).contiguous().to(device)
        self.boxes_h16 = boxes.to(dtype).to(device)              # pos_embedd input (layers.py:284-285) in the storage type
        self.boxes_bf16 = self.boxes_h16                         # (older name)
        off = [0]
        for c in self.counts:
            off.append(off[-1] + c)
        self.offsets = torch.tensor(off, dtype=torch.int32, device=device)
        self.image_size = image_size
        self.n = n


class BaseRoIExtractor(nn.Module):
    """mmdet BaseRoIExtractor.build_roi_layers (base_roi_extractor.py:37-60): one RoIAlign per
    stride, resolved here to gpt4roi_amd.roi_align.RoIAlign instead of `getattr(mmcv.ops, ...)`."""

    def __init__(self, roi_layer, out_channels, featmap_strides, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        layer_type = cfg.pop('type')
        assert layer_type == 'RoIAlign', layer_type
        self.roi_layers = nn.ModuleList([RoIAlign(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)


class MlvlRoIExtractor(BaseRoIExtractor):

    def __init__(self, roi_layer, out_channels, featmap_strides, embed_dims=1024, stride=1, norm_init=True,
                 fuse_level=3, finest_scale=56, init_cfg=None, out_dims=4096):
        super().__init__(roi_layer, out_channels, featmap_strides, init_cfg)
        self.embed_dims = embed_dims
        self.finest_scale = finest_scale      # stored, unused (as in the reference, layers.py:248)
        self.fuse_level = fuse_level
        self.norm_init = norm_init
        self.pconvs = nn.ModuleList(nn.Conv2d(self.embed_dims, self.embed_dims, 3, stride=1, padding=1)
                                    for _ in range(self.fuse_level))
        self.pos_embedd = nn.Sequential(nn.Linear(4, 256), nn.ReLU(inplace=True), nn.LayerNorm(256),
                                        nn.Linear(256, 1024), nn.ReLU(inplace=True), nn.LayerNorm(1024))
        self.updims = nn.Linear(1024, out_dims)
        self.flatten_linear = nn.Linear(self.embed_dims * self.roi_layers[0].output_size[0] ** 2, 1024)
        self.norm_init_weights()
        self._ready = None

    def norm_init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                normal_init(m, 0, 0.01)

    def prepare(self):
        bf = getattr(self, "compute_dtype", torch.bfloat16)
        C = self.embed_dims
        oh, ow = self.roi_layers[0].output_size
        with torch.no_grad():
            w_p = K.prep_conv3x3_weight([c.weight.detach() for c in self.pconvs], bf)
            b_p = sum(c.bias.detach().to(bf).float() for c in self.pconvs).contiguous()
            # flatten is c-major in the reference (layers.py:326): column c*oh*ow + pos.  Our RoI
            # features are NHWC, so permute the weight once: column pos*C + c.
            wf = self.flatten_linear.weight.detach()
            w_fl = wf.view(wf.size(0), C, oh * ow).permute(0, 2, 1).reshape(wf.size(0), oh * ow * C).to(bf).contiguous()

            def lin(l):
                return l.weight.detach().to(bf).contiguous(), l.bias.detach().to(bf).float().contiguous()

            def ln(l):
                return l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous(), l.eps
            self._ready = dict(w_p=w_p, b_p=b_p, w_fl=w_fl, b_fl=self.flatten_linear.bias.detach().to(bf).float().contiguous(),
                               pe0=lin(self.pos_embedd[0]), ln2=ln(self.pos_embedd[2]), pe3=lin(self.pos_embedd[3]),
                               ln5=ln(self.pos_embedd[5]), up=lin(self.updims))

    def forward(self, feats, rois, roi_scale_factor=None, affines=None, image_size=224):
        """feats: list of NHWC bf16 maps [B,H_l,W_l,C]; rois: list[B] of [n_i,4] normalised xyxy (or a
        PreparedBoxes built for `image_size`);
        affines: deferred GN+ReLU per level (from MLVLFuseModule.forward) or None."""
        if self._ready is None:
            self.prepare()
        r = self._ready
        dev = feats[0].device
        prep = rois if isinstance(rois, PreparedBoxes) else PreparedBoxes(rois, image_size, dev, dtype=feats[0].dtype)
        assert prep.dtype == feats[0].dtype, "PreparedBoxes were built for another storage type"
        num_imgs, counts, N = prep.num_imgs, prep.counts, prep.n
        out_dims = self.updims.out_features
        if N == 0:
            return [feats[0].new_zeros((0, out_dims)) for _ in range(num_imgs)]
        # pos_embedd(cat(bboxes)) on the raw normalised boxes (layers.py:284-285)
        pe = K.gemm(prep.boxes_h16, r['pe0'][0], bias=r['pe0'][1], act='relu')
        pe = K.layernorm(pe, r['ln2'][0], r['ln2'][1], r['ln2'][2])
        pe = K.gemm(pe, r['pe3'][0], bias=r['pe3'][1], act='relu')
        pe = K.layernorm(pe, r['ln5'][0], r['ln5'][1], r['ln5'][2])
        rois5 = prep.rois5                               # [img_id, box * image_size], layers.py:295-302
        rl = self.roi_layers[0]
        roi_feats = K.roi_align_mlvl(feats, rois5, rl.output_size, [l.spatial_scale for l in self.roi_layers],
                                     sampling_ratio=rl.sampling_ratio, aligned=rl.aligned, affines=affines)
        # sum_l pconv_l(roi_feats[l]) -> ReLU   (layers.py:321-325), one implicit GEMM
        fused = K.conv3x3(roi_feats, r['w_p'], bias=r['b_p'], act='relu', groups=self.fuse_level)
        flat = fused.view(N, -1)
        x = K.gemm(flat, r['w_fl'], bias=r['b_fl'], splits=64, tile_cfg=4)
        x = K.add_rows(x, pe)
        x = K.gemm(x, r['up'][0], bias=r['up'][1])
        return list(torch.split(x, counts, 0))


    # ---- training rows ---------------------------------------------------------------------------------------
    def forward_train(self, feats, rois, affines=None, image_size=224):
        """forward() that also returns the intermediates backward() needs."""
        if self._ready is None:
            self.prepare()
        r = self._ready
        dev = feats[0].device
        prep = rois if isinstance(rois, PreparedBoxes) else PreparedBoxes(rois, image_size, dev)
        assert prep.n > 0, "training needs at least one box"
        h1 = K.gemm(prep.boxes_h16, r['pe0'][0], bias=r['pe0'][1], act='relu')
        l2 = K.layernorm(h1, r['ln2'][0], r['ln2'][1], r['ln2'][2])
        h3 = K.gemm(l2, r['pe3'][0], bias=r['pe3'][1], act='relu')
        pe = K.layernorm(h3, r['ln5'][0], r['ln5'][1], r['ln5'][2])
        rl = self.roi_layers[0]
        roi_feats = K.roi_align_mlvl(feats, prep.rois5, rl.output_size, [l.spatial_scale for l in self.roi_layers],
                                     sampling_ratio=rl.sampling_ratio, aligned=rl.aligned, affines=affines)
        fused = K.conv3x3(roi_feats, r['w_p'], bias=r['b_p'], act='relu', groups=self.fuse_level)
        flat = fused.view(prep.n, -1)
        x = K.gemm(flat, r['w_fl'], bias=r['b_fl'], splits=64, tile_cfg=4)
        xs = K.add_rows(x, pe)
        out = K.gemm(xs, r['up'][0], bias=r['up'][1])
        ctx = dict(prep=prep, h1=h1, l2=l2, h3=h3, roi_feats=roi_feats, fused=fused, xs=xs,
                   shapes=[tuple(f.shape) for f in feats])
        return out, ctx

    def backward(self, ctx, d_out):
        """d_out bf16 [N, out_dims] -> ({state_dict key: fp32 gradient, reference layout}, list of fp32 NHWC
        gradients w.r.t. the (post GN+ReLU) feature maps)."""
        r = self._ready
        prep = ctx['prep']
        N, C, L = prep.n, self.embed_dims, self.fuse_level
        oh, ow = self.roi_layers[0].output_size
        dev = d_out.device
        d_out = d_out.contiguous()
        g = {}
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)  # noqa: E731
        # updims
        g['updims.weight'] = K.linear_wgrad(d_out, ctx['xs'])
        g['updims.bias'] = K.colsum(d_out)
        d_x = K.gemm(d_out, K.transpose(r['up'][0]))
        # pos_embedd: Linear -> ReLU -> LN -> Linear -> ReLU -> LN   (layers.py:260-267)
        g['pos_embedd.5.weight'], g['pos_embedd.5.bias'] = z(1024), z(1024)
        d_h3 = K.layernorm_bwd(ctx['h3'], r['ln5'][0], d_x, g['pos_embedd.5.weight'], g['pos_embedd.5.bias'], r['ln5'][2])
        d_p3 = K.relu_bwd(ctx['h3'], d_h3)
        g['pos_embedd.3.weight'] = K.linear_wgrad(d_p3, ctx['l2'])
        g['pos_embedd.3.bias'] = K.colsum(d_p3)
        d_l2 = K.gemm(d_p3, K.transpose(r['pe3'][0]))
        g['pos_embedd.2.weight'], g['pos_embedd.2.bias'] = z(256), z(256)
        d_h1 = K.layernorm_bwd(ctx['h1'], r['ln2'][0], d_l2, g['pos_embedd.2.weight'], g['pos_embedd.2.bias'], r['ln2'][2])
        d_p1 = K.relu_bwd(ctx['h1'], d_h1)
        g['pos_embedd.0.weight'] = K.linear_wgrad(d_p1, prep.boxes_h16)
        g['pos_embedd.0.bias'] = K.colsum(d_p1)
        # flatten_linear (kernel layout: column pos*C + c; reference: column c*oh*ow + pos)
        flat = ctx['fused'].view(N, -1)
        dwf = K.linear_wgrad(d_x, flat)
        g['flatten_linear.weight'] = dwf.view(-1, oh * ow, C).permute(0, 2, 1).reshape(dwf.size(0), -1).contiguous()
        g['flatten_linear.bias'] = K.colsum(d_x)
        d_flat = K.gemm(d_x, K.transpose(r['w_fl']))
        d_pre = K.relu_bwd(flat, d_flat).view(N, oh, ow, C)
        # pconvs: sum_l conv_l(roi_feats[l]) + sum_l bias_l
        db = K.colsum(d_pre.view(-1, C))
        if getattr(self, '_pplan', None) is None or self._pplan.B != N:
            self._pplan = K.ConvWgradPlan(N, oh, ow, C, C, dev)
        for l in range(L):
            g[f'pconvs.{l}.bias'] = db if l == 0 else db.clone()
            g[f'pconvs.{l}.weight'] = self._pplan.wgrad(ctx['roi_feats'][l], d_pre)
        d_roi = K.conv3x3(d_pre, K.conv3x3_dgrad_weight([c.weight.detach() for c in self.pconvs]))  # [N,oh,ow,L*C]
        # the gather kernel writes every texel once: no zero-fill, no atomics, fixed summation order
        d_maps = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx['shapes']]
        rl = self.roi_layers[0]
        K.roi_align_mlvl_bwd(d_roi, C, L * C, d_maps, prep.rois5, rl.output_size,
                             [l.spatial_scale for l in self.roi_layers], rl.sampling_ratio, rl.aligned,
                             roi_offsets=prep.offsets)
        return g, d_maps


class MLVLROIQueryModule(nn.Module):

    def __init__(self, embed_dims=1024, out_dims=4096, num_levels=3):
        super().__init__()
        self.mlvl_fuse = MLVLFuseModule(input_dims=embed_dims, embed_dims=embed_dims, num_levels=num_levels,
                                        num_fuse=5)
        strids = [14 / 8, 14 / 4, 14 / 2, 14]
        assert len(strids) == num_levels
        bbox_roi_extractor = dict(roi_layer=dict(type='RoIAlign', output_size=14, sampling_ratio=2),
                                  out_channels=embed_dims, embed_dims=embed_dims, fuse_level=num_levels,
                                  featmap_strides=strids, out_dims=out_dims)
        self.roi_align = MlvlRoIExtractor(**bbox_roi_extractor)

    compute_dtype = torch.bfloat16

    def set_compute_dtype(self, dtype):
        """16-bit storage type of the kernel-ready weights and every activation of the module: torch.bfloat16 (default, the
        training dtype) or torch.float16 (the reference's serving dtype, app.py:74-98).  Inference only for fp16."""
        assert dtype in K.H16
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self.mlvl_fuse._ready = self.roi_align._ready = None
        return self

    def prepare(self):
        self.mlvl_fuse.compute_dtype = self.roi_align.compute_dtype = self.compute_dtype
        self.mlvl_fuse.prepare()
        self.roi_align.prepare()
        self._stamp = self._param_stamp()

    def _tokens(self, mlvl_feats):
        toks = []
        for f in mlvl_feats:
            if f.dim() == 4:                                   # NCHW -> token form
                b, c, h, w = f.shape
                f = f.permute(0, 2, 3, 1).reshape(b, h * w, c)
            if f.dtype != self.compute_dtype:
                f = f.to(self.compute_dtype)
            toks.append(f)
        P = int(math.isqrt(toks[0].shape[1]))
        assert P * P == toks[0].shape[1], "level features must be square token grids"
        n = len(toks)
        sizes = [P * 2 ** l for l in range(n)][::-1]          # level 0 (shallowest ViT layer) is finest
        return toks, P, sizes

    @torch.no_grad()
    def forward_train(self, mlvl_feats, bboxes):
        """Training forward: (region embeddings [N, out_dims] bf16 concatenated over the batch, ctx)."""
        toks, P, sizes = self._tokens(mlvl_feats)
        maps, affs, fctx = self.mlvl_fuse.forward_train(toks, P, sizes)
        out, rctx = self.roi_align.forward_train(maps, bboxes, affines=affs, image_size=14 * P)
        return out, dict(fuse=fctx, roi=rctx)

    @torch.no_grad()
    def backward(self, ctx, d_out, on_grad=None):
        """d_out bf16 [N, out_dims] (rows in the order of cat(bboxes)) -> {state_dict key: fp32 gradient in the
        reference layout} for every parameter of the module (the gradients autograd gives the reference through
        gpt4roi/models/layers.py:218-236).  `on_grad(key, grad)` fires as gradients complete, in the reverse of the
        module's parameter registration order (roi_align.* first, then the fuse rounds from the last to the first)."""
        g_roi, d_maps = self.roi_align.backward(ctx['roi'], d_out)
        grads = {f'roi_align.{k}': v for k, v in g_roi.items()}
        if on_grad is not None:
            for k in reversed([n for n, _ in self.roi_align.named_parameters()]):
                on_grad(f'roi_align.{k}', grads[f'roi_align.{k}'])
        cb = (lambda k, g: on_grad(f'mlvl_fuse.{k}', g)) if on_grad is not None else None
        g_fuse = self.mlvl_fuse.backward(ctx['fuse'], d_maps, on_grad=cb)
        grads.update({f'mlvl_fuse.{k}': v for k, v in g_fuse.items()})
        return grads

    def _param_stamp(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _maybe_prepare(self):
        """Re-derive the kernel-ready bf16 buffers when a parameter was written since the last prepare() (an
        optimizer step, load_state_dict, .to()): tensors carry a version counter that every in-place write bumps."""
        if self.mlvl_fuse._ready is None or self.roi_align._ready is None or \
                getattr(self, '_stamp', None) != self._param_stamp():
            self.prepare()
            self._stamp = self._param_stamp()

    def forward(self, mlvl_feats, bboxes):
        """mlvl_feats: list[4] of [B, P*P, C] (token form, as spi_llava.py:80-82 passes) or
        [B, C, P, P]; bboxes: list[B] of [n_i, 4] normalised xyxy.  Returns list[B] of
        [n_i, out_dims] bf16.  With grad enabled and trainable parameters the result carries an autograd node whose
        backward is the hand-written `backward` above (gradients for the module's parameters; the ViT features are
        treated as constants, as the reference computes them under no_grad, spi_llava.py:50-82)."""
        self._maybe_prepare()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            n_total = bboxes.n if isinstance(bboxes, PreparedBoxes) else sum(int(b.size(0)) for b in bboxes)
            if n_total > 0:
                named = [(k, p) for k, p in self.named_parameters() if p.requires_grad]
                out = _RegionModuleFn.apply(self, mlvl_feats, bboxes, tuple(k for k, _ in named), *[p for _, p in named])
                counts = bboxes.counts if isinstance(bboxes, PreparedBoxes) else [int(b.size(0)) for b in bboxes]
                return list(torch.split(out, counts, 0))
        with torch.no_grad():
            toks, P, sizes = self._tokens(mlvl_feats)
            maps, affs = self.mlvl_fuse(toks, P, sizes)
            if isinstance(bboxes, PreparedBoxes):
                assert bboxes.image_size == 14 * P, "PreparedBoxes built for another image size"
            return self.roi_align(maps, bboxes, affines=affs, image_size=14 * P)


class _RegionModuleFn(torch.autograd.Function):
    """B2 seam under autograd: out = MLVLROIQueryModule(mlvl_feats, bboxes; parameters)."""

    @staticmethod
    def forward(ctx, module, mlvl_feats, bboxes, names, *params):
        with torch.no_grad():
            out, sctx = module.forward_train(mlvl_feats, bboxes)
        ctx.module, ctx.sctx, ctx.names = module, sctx, names
        return out

    @staticmethod
    def backward(ctx, d_out):
        with torch.no_grad():
            grads = ctx.module.backward(ctx.sctx, d_out.to(torch.bfloat16).contiguous())
        ctx.sctx = None
        return (None, None, None, None) + tuple(grads[n] for n in ctx.names)
