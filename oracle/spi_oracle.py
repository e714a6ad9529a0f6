"""oracle/spi_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain torch, fp32) of the reference's region module and token splice:

  MLVLFuseModule        /root/reference/gpt4roi/models/layers.py:96-195
  MLVLROIQueryModule    layers.py:198-236
  MlvlRoIExtractor      layers.py:239-335
  ConvModule            mmcv-1.4.7/mmcv/cnn/bricks/conv_module.py:104-105,196-206 (conv, bias off
                        when a norm follows -> GN(64, eps 1e-5, attribute `gn`) -> ReLU)
  splice / <bbox> inject gpt4roi/models/spi_llava.py:99-196

with the reference's state_dict key names (`mlvl_fuse.input_conv.N`,
`mlvl_fuse.fuse_convs.N.conv|gn`, `roi_align.pconvs.N`, `roi_align.pos_embedd.{0,2,3,5}`,
`roi_align.updims`, `roi_align.flatten_linear`).  RoIAlign is the C oracle
(oracle/roi_align_oracle.c, itself pinned to the reference's compiled CPU code).

Pinning: the reference has NO test for this module.  tests/golden/make_spi_golden.py imports the
reference's own gpt4roi/models/layers.py in the build container (its mmcv/mmdet leaf imports
replaced by the restated ConvModule/BaseRoIExtractor of this file) and commits its outputs as
tests/golden/spi_module_ref.npz; tests/test_oracle_spi.py checks this restatement against them.

Differences from the reference, all deliberate and parameterised:
  * the hard-wired 224/16 constants (layers.py:220-222, 289-291, 297) become `P = image_size/14`;
    at P = 16 the behaviour is the reference's;
  * `emulate_bf16=True` rounds to bfloat16 at the points where the MI355X pipeline stores bf16
    (DESIGN.md "rounding points"), so kernel bugs are not hidden under bf16 noise.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import roi_align as roi_oracle


def _r(x, emulate):
    """emulate: False = exact fp32; True = round to bfloat16 (the training dtype); a torch dtype (torch.float16: the
    reference's serving dtype, app.py:74-98) = round to that type."""
    if not emulate:
        return x
    return x.to(torch.bfloat16 if emulate is True else emulate).to(torch.float32)


class ConvModuleOracle(nn.Module):
    """mmcv ConvModule(3x3, norm_cfg=GN/64): conv(bias=False) -> gn -> relu."""

    def __init__(self, cin, cout, groups=64):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=False)
        self.gn = nn.GroupNorm(groups, cout, eps=1e-5)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')

    def forward(self, x, emulate=False):
        y = F.conv2d(_r(x, emulate), _r(self.conv.weight, emulate), None, padding=1)
        return F.relu(self.gn(_r(y, emulate)))


class _RoIAlignOracleFn(torch.autograd.Function):
    """The C oracle's forward and backward (oracle/roi_align_oracle.c, both pinned bit-exact to the reference's CPU
    build) as one autograd node: what mmcv's RoIAlignFunction is for the reference (mmcv/ops/roi_align.py:64-128)."""

    @staticmethod
    def forward(ctx, x, rois, output_size, spatial_scale, sampling_ratio, pool_mode, aligned):
        assert pool_mode == 'avg'
        # (x may live on an accelerator when the surrounding torch ops of the oracle are run there: the node itself is always
        #  the C oracle on the host)
        out, _, _ = roi_oracle.forward(x.detach().float().cpu().numpy(), rois.detach().float().cpu().numpy(), output_size,
                                       np.float32(spatial_scale), sampling_ratio, pool_mode, aligned)
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(x.shape), output_size, spatial_scale, sampling_ratio, pool_mode, aligned)
        return torch.from_numpy(out).to(x.device)

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        shape, output_size, spatial_scale, sampling_ratio, pool_mode, aligned = ctx.cfg
        gin = roi_oracle.backward(grad_out.contiguous().float().cpu().numpy(), rois.detach().float().cpu().numpy(), shape,
                                  output_size, np.float32(spatial_scale), sampling_ratio, pool_mode, aligned)
        return torch.from_numpy(gin).to(grad_out.device), None, None, None, None, None, None


class RoIAlignOracle(nn.Module):
    """mmcv.ops.RoIAlign restated on the C oracle (differentiable w.r.t. the feature map)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned

    def forward(self, x, rois):
        if x.requires_grad:
            return _RoIAlignOracleFn.apply(x, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                                           self.pool_mode, self.aligned)
        out, _, _ = roi_oracle.forward(x.detach().float().cpu().numpy(), rois.detach().float().cpu().numpy(),
                                       self.output_size, np.float32(self.spatial_scale), self.sampling_ratio,
                                       self.pool_mode, self.aligned)
        return torch.from_numpy(out).to(x.device)


class BaseRoIExtractorOracle(nn.Module):
    """mmdet BaseRoIExtractor.build_roi_layers (base_roi_extractor.py:37-60)."""

    def __init__(self, roi_layer, out_channels, featmap_strides, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        assert cfg.pop('type') == 'RoIAlign'
        self.roi_layers = nn.ModuleList([RoIAlignOracle(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides


class MLVLFuseOracle(nn.Module):

    def __init__(self, input_dims=1024, embed_dims=1024, num_levels=4, num_fuse=5):
        super().__init__()
        self.embed_dims, self.num_levels = embed_dims, num_levels
        self.shuffle = embed_dims // 4
        self.remain = embed_dims - 2 * self.shuffle
        self.fuse_lvl_list = [(l, min(l + 1, num_levels - 1), max(l - 1, 0)) for l in range(num_levels)]
        self.input_conv = nn.ModuleList([nn.Conv2d(input_dims + 2, embed_dims, 1) for _ in range(num_levels)])
        self.fuse_convs = nn.ModuleList([ConvModuleOracle(embed_dims, embed_dims) for _ in range(num_fuse)])

    @staticmethod
    def coords(shape):
        x_range = torch.linspace(-1, 1, shape[-1])
        y_range = torch.linspace(-1, 1, shape[-2])
        y, x = torch.meshgrid(y_range, x_range, indexing='ij')
        y = y.expand([shape[0], 1, -1, -1])
        x = x.expand([shape[0], 1, -1, -1])
        return torch.cat([x, y], 1)

    def forward(self, inputs, emulate=False):
        xs = []
        for lvl, feat in enumerate(inputs):
            feat = torch.cat([feat, _r(self.coords(feat.shape).to(feat.device), emulate)], 1)
            conv = self.input_conv[lvl]
            y = F.conv2d(_r(feat, emulate), _r(conv.weight, emulate), _r(conv.bias, emulate))
            xs.append(_r(y, emulate))
        for m in self.fuse_convs:
            fused = []
            for tar, top, dow in self.fuse_lvl_list:
                t = xs[tar]
                size = t.shape[-2:]
                from_top = F.interpolate(xs[top][:, self.remain:][:, self.shuffle:].float(), size=size,
                                         mode='bilinear', align_corners=True)
                from_down = F.interpolate(xs[dow][:, self.remain:][:, :self.shuffle].float(), size=size,
                                          mode='bilinear', align_corners=True)
                fused.append(torch.cat([t[:, :self.remain], from_top, from_down], 1))
            xs = [m(item, emulate) for item in fused]
        return xs


class MlvlRoIExtractorOracle(BaseRoIExtractorOracle):

    def __init__(self, roi_layer, out_channels, featmap_strides, embed_dims=1024, fuse_level=4, image_size=224):
        super().__init__(roi_layer, out_channels, featmap_strides)
        self.embed_dims, self.fuse_level, self.image_size = embed_dims, fuse_level, image_size
        self.pconvs = nn.ModuleList(nn.Conv2d(embed_dims, embed_dims, 3, stride=1, padding=1)
                                    for _ in range(fuse_level))
        self.pos_embedd = nn.Sequential(nn.Linear(4, 256), nn.ReLU(inplace=True), nn.LayerNorm(256),
                                        nn.Linear(256, 1024), nn.ReLU(inplace=True), nn.LayerNorm(1024))
        self.updims = nn.Linear(1024, 4096)
        self.flatten_linear = nn.Linear(embed_dims * self.roi_layers[0].output_size[0] ** 2, 1024)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _lin(self, layer, x, emulate):
        return _r(F.linear(_r(x, emulate), _r(layer.weight, emulate), _r(layer.bias, emulate)), emulate)

    def forward(self, feats, rois, emulate=False):
        num_imgs = len(rois)
        batch_rois = torch.cat(rois, 0).float()
        pe = batch_rois
        pe = F.layer_norm(F.relu(self._lin(self.pos_embedd[0], pe, emulate)), (256,), self.pos_embedd[2].weight,
                          self.pos_embedd[2].bias, 1e-5)
        pe = F.layer_norm(F.relu(self._lin(self.pos_embedd[3], pe, emulate)), (1024,), self.pos_embedd[5].weight,
                          self.pos_embedd[5].bias, 1e-5)
        pe = _r(pe, emulate)
        new_rois = []
        for img_id, r in enumerate(rois):
            if emulate is torch.float16:
                # serving: the boxes arrive as .half() (app.py:271), so `single_img_roi * 224` (layers.py:297) is a half
                # product, rounded to fp16 before RoIAlign's .to(float32) (layers.py:311)
                r = (r.to(torch.float16) * self.image_size).float()
            else:
                r = r.float() * self.image_size                  # layers.py:297 (224 at P = 16)
            new_rois.append(torch.cat([r.new_ones(len(r), 1) * img_id, r], 1))
        rois5 = torch.cat(new_rois)
        roi_feats = [self.roi_layers[i](feats[i].float(), rois5) for i in range(len(feats))]
        acc = 0
        for i in range(self.fuse_level):
            c = self.pconvs[i]
            acc = acc + F.conv2d(_r(roi_feats[i], emulate), _r(c.weight, emulate), _r(c.bias, emulate), padding=1)
        x = _r(F.relu(acc), emulate).flatten(1, -1)              # c-major flatten, layers.py:326
        x = self._lin(self.flatten_linear, x, emulate)
        x = _r(x + pe, emulate)
        x = self._lin(self.updims, x, emulate)
        return [x[rois5[:, 0] == i] for i in range(num_imgs)], roi_feats


class MLVLROIQueryOracle(nn.Module):

    def __init__(self, embed_dims=1024, out_dims=4096, num_levels=4, P=16):
        super().__init__()
        self.P = P
        self.mlvl_fuse = MLVLFuseOracle(embed_dims, embed_dims, num_levels, num_fuse=5)
        strides = [14 / 8, 14 / 4, 14 / 2, 14]
        self.roi_align = MlvlRoIExtractorOracle(dict(type='RoIAlign', output_size=14, sampling_ratio=2),
                                                embed_dims, strides, embed_dims=embed_dims, fuse_level=num_levels,
                                                image_size=14 * P)

    def pyramid(self, mlvl_feats, emulate=False):
        if mlvl_feats[0].dim() == 3:
            h = w = int(math.sqrt(mlvl_feats[0].shape[1]))
            assert h == self.P
            b, c = mlvl_feats[0].shape[0], mlvl_feats[0].shape[-1]
            mlvl_feats = [f.reshape(b, h, w, c).permute(0, 3, 1, 2) for f in mlvl_feats]
        base = mlvl_feats[0].shape[-2:]
        n = len(mlvl_feats)
        to_shape = [(base[0] * 2 ** l, base[1] * 2 ** l) for l in range(n)][::-1]
        return [_r(F.interpolate(mlvl_feats[l].float(), size=to_shape[l], mode='bilinear', align_corners=True),
                   emulate) for l in range(n)]

    def forward(self, mlvl_feats, bboxes, emulate=False, return_intermediates=False):
        pyr = self.pyramid(mlvl_feats, emulate)
        fused = self.mlvl_fuse(pyr, emulate)
        out, roi_feats = self.roi_align(fused, bboxes, emulate)
        if return_intermediates:
            return out, dict(pyramid=pyr, fused=fused, roi_feats=roi_feats)
        return out


def synthetic_state(module, seed):
    """Deterministic weights independent of module-construction order: every tensor of the
    state_dict, in sorted key order, from one seeded generator (std chosen per kind so that
    activations stay O(1) through 5 fuse rounds)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(module.state_dict().keys()):
        v = module.state_dict()[k]
        if k.endswith('gn.weight') or (k.endswith('.weight') and v.dim() == 1):
            t = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith('.bias'):
            t = 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            t = torch.randn(v.shape, generator=g) * (1.5 / math.sqrt(fan_in))
        sd[k] = t.to(v.dtype)
    return sd


def synthetic_inputs(seed, B, P, C, n_rois):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(B, P * P, C, generator=g) for _ in range(4)]
    boxes = []
    for n in n_rois:
        xy = torch.rand(n, 2, generator=g) * 0.6
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.05
        boxes.append(torch.cat([xy, xy + wh], 1))
    return feats, boxes


# ---- token splice (gpt4roi/models/spi_llava.py:99-196), use_im_start_end=True branch ----
def splice(input_ids, inputs_embeds, image_features, spi_feats, im_start, im_end, bbox_id):
    out = []
    for ids, emb, img, spi in zip(input_ids, inputs_embeds, image_features, spi_feats):
        starts = torch.where(ids == im_start)[0]
        if (ids == im_start).sum() != (ids == im_end).sum():
            raise ValueError('The number of image start tokens and image end tokens should be the same.')
        cur = emb
        for s in starts:
            n = img.shape[0]
            if ids[s + n + 1] != im_end:
                raise ValueError('The image end token should follow the image start token.')
            cur = torch.cat((cur[:s + 1], img, cur[s + n + 1:]), 0)
            if spi is not None:
                mask = ids == bbox_id
                se = torch.zeros_like(cur)
                se[mask] = spi.to(se.dtype)
                cur = cur * (~mask).to(cur.dtype)[:, None] + se
            else:
                assert (ids == bbox_id).sum() == 0
        out.append(cur)
    return torch.stack(out, 0)
