"""Drop-in for the reference's models/cpn.py: masked_softmax, the two context-matching modules, the border helpers,
the affine alignment warp of CPNet.align, and CPNet itself (same constructor, sub-module names and state_dict keys, so a
checkpoint of the reference loads unchanged).  The conv encoders / regressor / decoder are stock torch convolutions (they are
not on the hand-written path); `align` and `copy_and_paste` run the HIP affine warp and the HIP context matching."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, _ops


def masked_softmax(vec, mask, dim):
    """Softmax of `vec` over `dim` restricted to mask == 1 (zero-safe).  Mirrors models/cpn.py:7-15."""
    return _ops.MaskedSoftmaxFn.apply(vec, mask, dim)


def add_match_border(v_t, c_match):
    """ref_importance of the default variant: c_match[:, 0] (B,R,h,w) upsampled to v_t's size, a zero plane inserted in
    the middle, the hole border painted red -> (B,3,R+1,H,W).  Mirrors models/cpn.py:23-36, on the device."""
    return _ops.ref_importance(c_match[:, 0], v_t, middle_is_vt=False)


def add_border(x, mask):
    """Paint the outer border of `mask` (B,1,H,W) red on x (B,3,F,H,W).  Mirrors models/cpn.py:39-44, on the device."""
    border = ((_ops.dilate_cross3(mask) - mask) > 0).float().unsqueeze(2)
    red = torch.tensor([1.0, 0.0, 0.0], device=x.device).view(1, 3, 1, 1, 1)
    return x * (1 - border) + border * red


class _ContextMatching(nn.Module):
    _variant = _lib.CTX_DEFAULT
    _middle_is_vt = False

    def forward(self, c_feats, v_t, v_aligned, normalize=True):
        b, c_c, f, h, w = c_feats.size()
        p_in, c_mask, c_match = _ops.ContextMatchingFn.apply(c_feats, v_t, v_aligned, normalize, self._variant)
        ref_importance = _ops.ref_importance(c_match, v_t, self._middle_is_vt)
        return p_in, c_mask, ref_importance


class CPNContextMatchingDefault(_ContextMatching):
    """(c_feats (B,C,R+1,h,w), v_t (B,1,H,W), v_aligned (B,1,R,H,W), normalize) ->
    (cat[target feats, aggregated refs, c_mask] (B,2C+1,h,w), c_mask (B,1,h,w), ref_importance (B,3,R+1,H,W)).
    Mirrors models/cpn.py:90-135."""
    _variant = _lib.CTX_DEFAULT
    _middle_is_vt = False


class CPNContextMatchingComplete(_ContextMatching):
    """Hole-only variant -> (cat[target*v_t + aggregated refs, c_mask] (B,C+1,h,w), c_mask, ref_importance).
    Mirrors models/cpn.py:138-195."""
    _variant = _lib.CTX_COMPLETE
    _middle_is_vt = True


def affine_align(x, m, y, theta_rt, r_list, training=False):
    """The affine_grid + grid_sample block of CPNet.align (models/cpn.py:280-295) / CPNOriginal.align
    (models/cpn_original.py:276-284): theta_rt (B*R,2,3) -> x_aligned (B,C,R,H,W), v_aligned (B,1,R,H,W) in {0,1},
    y_aligned | None.  The sampling grid is derived from theta inside the kernel and never stored."""
    # the reference gathers x[:, :, r_list] / m[:, :, r_list] first (an index copy as long as the warp itself); here the warp
    # reads the reference frames in place (WarpFn's `frames`)
    frames = [int(i) for i in r_list]
    # ... and flips the hole masks per tap (v_is_hole): no `1 - m` over all R + 1 frames either
    x_aligned, v_aligned = _ops.WarpFn.apply(x, m, None, theta_rt, False, _lib.VMODE_BILINEAR_THRESHOLD, frames, True)
    y_aligned = None
    if not training and y is not None:
        y_aligned, _ = _ops.WarpFn.apply(y, None, None, theta_rt, False, _lib.VMODE_NONE, frames)
    return x_aligned, v_aligned, y_aligned


# ---------------------------------------------------------------------------------------------- CPNet (boundary)
def _convs(spec, in_c, fan_out_init=False):
    """spec rows: (out_c, kernel, stride, dilation, relu) -> nn.Sequential of Conv2d [+ ReLU] with 'same'-style padding.
    Module indices (and so the state_dict keys `convs.<i>.weight`) match the reference's stacks."""
    layers, c = [], in_c
    for out_c, k, stride, dil, relu in spec:
        layers.append(nn.Conv2d(c, out_c, k, stride=stride, padding=dil * (k // 2), dilation=dil))
        if fan_out_init:   # models/cpn.py:18-20
            nn.init.kaiming_normal_(layers[-1].weight, mode='fan_out', nonlinearity='relu')
        if relu:
            layers.append(nn.ReLU())
        c = out_c
    return nn.Sequential(*layers)


class CPNAlignmentEncoder(nn.Module):
    """(frames (N,3,h,w), visibility (N,1,h,w)) -> (N,256,28,28).  models/cpn.py:47-64."""

    def __init__(self):
        super().__init__()
        self.convs = _convs([(64, 5, 2, 1, 1), (64, 3, 1, 1, 1), (128, 3, 2, 1, 1), (128, 3, 1, 1, 1),
                             (256, 3, 2, 1, 1), (256, 3, 1, 1, 1), (256, 3, 1, 1, 1)], 4, fan_out_init=True)

    def forward(self, in_f, in_v):
        return self.convs(F.interpolate(torch.cat([in_f, in_v], dim=1), size=(224, 224), mode='bilinear', align_corners=False))


class CPNAlignmentRegressor(nn.Module):
    """(reference feats, target feats) -> theta (N,2,3), identity at initialisation.  models/cpn.py:67-87."""

    def __init__(self):
        super().__init__()
        self.convs = _convs([(512, 3, 2, 1, 1), (512, 3, 1, 1, 1), (512, 3, 1, 1, 1)] * 2, 512, fan_out_init=True)
        self.fc = nn.Linear(512, 6)
        with torch.no_grad():
            self.fc.bias.copy_(torch.tensor([1., 0., 0., 0., 1., 0.]))

    def forward(self, feat1, feat2):
        x = self.convs(torch.cat([feat1, feat2], dim=1))
        return self.fc(F.avg_pool2d(x, x.shape[2]).flatten(1)).view(-1, 2, 3)


class CPNEncoderDefault(nn.Module):
    """stride-8 copy encoder, 4 -> 128 channels.  models/cpn.py:198-212."""

    def __init__(self, input_channels=4):
        super().__init__()
        self.convs = _convs([(64, 5, 2, 1, 1), (64, 3, 1, 1, 1), (128, 3, 4, 1, 1), (128, 3, 1, 1, 1), (128, 3, 1, 1, 0)], input_channels)

    def forward(self, x, v):
        return self.convs(torch.cat([x, v], dim=1)), None


class CPNDecoderDefault(nn.Module):
    """257 -> 3 channels with the dilated block, x4 and x2 nearest upsampling.  models/cpn.py:215-241."""

    def __init__(self, in_c=257):
        super().__init__()
        self.convs = _convs([(257, 3, 1, 1, 1)] * 3 + [(257, 3, 1, d, 1) for d in (2, 4, 8, 16)] +
                            [(257, 3, 1, 1, 1), (128, 3, 1, 1, 1), (128, 3, 1, 1, 1)], in_c)
        self.convs_2 = _convs([(64, 3, 1, 1, 1), (64, 3, 1, 1, 1)], 128)
        self.convs_3 = nn.Conv2d(64, 3, 5, stride=1, padding=2)

    def forward(self, x, c_feats_mid=None):
        x = self.convs_2(F.interpolate(self.convs(x), scale_factor=4, mode='nearest'))
        return self.convs_3(F.interpolate(x, scale_factor=2, mode='nearest'))


class CPNet(nn.Module):
    """models/cpn.py:244-363 around the HIP kernels.  x, y (B,3,F,H,W), m (B,1,F,H,W) with 1 = hole, t the target index,
    r_list the reference indices."""
    _modes_all = ['full', 'aligner', 'encdec']

    def __init__(self, mode, utils_alignment=None):
        super().__init__()
        assert mode in self._modes_all
        self.mode = mode
        self.utils_alignment = utils_alignment
        if utils_alignment is None and mode in ['full', 'aligner']:
            self.alignment_encoder = CPNAlignmentEncoder()
            self.alignment_regressor = CPNAlignmentRegressor()
        if mode in ['full', 'encdec']:
            self.context_matching = CPNContextMatchingDefault()
            self.encoder = CPNEncoderDefault()
            self.decoder = CPNDecoderDefault()
        self.register_buffer('mean', torch.as_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1))
        self.register_buffer('std', torch.as_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1))

    def align(self, x, m, y, t, r_list):
        """-> (x_aligned (B,3,R,H,W), v_aligned (B,1,R,H,W) in {0,1}, y_aligned | None when training).  models/cpn.py:266-298;
        theta -> sampling positions inside the warp kernel, x and the visibility share one launch."""
        b, c, f, h, w = x.size()
        feats = self.alignment_encoder(x.transpose(1, 2).reshape(-1, c, h, w), m.transpose(1, 2).reshape(-1, 1, h, w))
        feats = feats.reshape(b, f, *feats.shape[1:])
        refs = feats[:, r_list].reshape(-1, *feats.shape[2:])
        tgt = feats[:, t].unsqueeze(1).expand(b, len(r_list), *feats.shape[2:]).reshape(-1, *feats.shape[2:])
        theta_rt = self.alignment_regressor(refs, tgt)
        return affine_align(x, m, y, theta_rt.float(), r_list, training=self.training)

    def copy_and_paste(self, x_t, m_t, y_t, x_aligned, v_aligned):
        """-> (y_hat, y_hat_comp, c_mask (B,1,H,W), ref_importance).  models/cpn.py:300-331."""
        b, c, f_ref, h, w = x_aligned.size()
        frames = torch.cat([x_t.unsqueeze(2), x_aligned], dim=2).transpose(1, 2).reshape(-1, c, h, w)
        vis = torch.cat([1 - m_t.unsqueeze(2), v_aligned], dim=2).transpose(1, 2).reshape(-1, 1, h, w)
        c_feats, c_feats_mid = self.encoder(frames, vis)
        c_feats = c_feats.reshape(b, f_ref + 1, *c_feats.shape[1:]).transpose(1, 2)
        if c_feats_mid is not None:
            c_feats_mid = [cm.reshape(b, f_ref + 1, *cm.shape[1:]).mean(dim=1) for cm in c_feats_mid]
        p_in, c_mask, ref_importance = self.context_matching(c_feats, 1 - m_t, v_aligned, normalize=False)
        c_mask = F.interpolate(c_mask, size=(h, w), mode='bilinear', align_corners=False).detach()
        y_hat = torch.clamp(self.decoder(p_in, c_feats_mid) * self.std.squeeze(4) + self.mean.squeeze(4), 0, 1)
        return y_hat, y_hat * m_t + y_t * (1 - m_t), c_mask, ref_importance

    def forward(self, x, m, y, t, r_list):
        if self.utils_alignment is None:
            x = (x - self.mean) / self.std
            x_aligned, v_aligned, _ = self.align(x, m, y, t, r_list)
        elif self.utils_alignment.model_name == 'cpn':
            x = (x - self.mean) / self.std
            x_aligned, v_aligned, _ = self.utils_alignment.align(x, m, y, t, r_list)
        else:
            x_aligned, v_aligned, _ = self.utils_alignment.align(x, m, y, t, r_list)
            x_aligned = (x_aligned - self.mean) / self.std
        if self.mode == 'full':
            y_hat, y_hat_comp, c_mask, ref_importance = self.copy_and_paste(x[:, :, t], m[:, :, t], y[:, :, t], x_aligned, v_aligned)
        elif self.mode == 'encdec':
            # (kept as the reference has it: this branch hands the encoder's (features, None) tuple to the decoder and so
            # fails there, models/cpn.py:350-351; nothing in the reference selects it)
            c_feats = self.encoder(x[:, :, t], 1 - m[:, :, t])
            y_hat = torch.clamp(self.decoder(c_feats) * self.std.squeeze(4) + self.mean.squeeze(4), 0, 1)
            return y_hat, y_hat * m[:, :, t] + y[:, :, t] * (1 - m[:, :, t]), m.squeeze(2), (x, 1 - m)
        else:
            y_hat, y_hat_comp, c_mask, ref_importance = None, None, None, None
        x_aligned = x_aligned * self.std + self.mean if x_aligned is not None else x_aligned
        return y_hat, y_hat_comp, c_mask, ref_importance, (x_aligned, v_aligned)
