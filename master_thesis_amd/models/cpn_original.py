"""Drop-in for the context-matching module of the reference's models/cpn_original.py."""
import torch.nn as nn

from .. import _lib, _ops


class CM_Module(nn.Module):
    """(values (B,C,T+1,h,w), tvmap (B,1,H,W), rvmaps (B,1,T,H,W)) -> (cat[t_feat, c_out, c_mask], c_mask).
    Mirrors models/cpn_original.py:187-243 (similarity always divided by the overlap area and by C)."""

    def masked_softmax(self, vec, mask, dim):
        return _ops.MaskedSoftmaxFn.apply(vec, mask, dim)

    def forward(self, values, tvmap, rvmaps):
        B, C, T, H, W = values.size()
        p_in, c_mask, _ = _ops.ContextMatchingFn.apply(values, tvmap, rvmaps, True, _lib.CTX_CM_MODULE)
        return p_in, c_mask


# ---------------------------------------------------------------------------------------------- the whole network (C4)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from . import cpn as _cpn  # noqa: E402


class Conv2d(nn.Module):
    """Convolution [+ activation] under the attribute `conv` (state_dict keys `<name>.conv.0.weight`).  models/cpn_original.py:6-21."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, D=1, activation=nn.ReLU()):
        super().__init__()
        layers = [nn.Conv2d(in_ch, out_ch, kernel_size=kernel_size, stride=stride, padding=padding, dilation=D)]
        self.conv = nn.Sequential(*(layers + [activation] if activation else layers))

    def forward(self, x):
        return self.conv(x)


def init_He(module):
    """models/cpn_original.py:24-30"""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def _named_convs(owner, rows):
    """rows: (attribute name, in_c, out_c, kernel, stride, dilation, relu); registers them on `owner` in order."""
    for name, cin, cout, k, stride, dil, relu in rows:
        setattr(owner, name, Conv2d(cin, cout, kernel_size=k, stride=stride, padding=dil * (k // 2), D=dil, activation=nn.ReLU() if relu else None))


def _imagenet_buffers(owner):
    owner.register_buffer('mean', torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
    owner.register_buffer('std', torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))


class A_Encoder(nn.Module):
    """Alignment encoder on 224 x 224 resized inputs -> (N,256,28,28).  models/cpn_original.py:52-77."""
    _order = ['conv12', 'conv2', 'conv23', 'conv3', 'conv34', 'conv4a', 'conv4b']

    def __init__(self):
        super().__init__()
        _named_convs(self, [('conv12', 4, 64, 5, 2, 1, 1), ('conv2', 64, 64, 3, 1, 1, 1), ('conv23', 64, 128, 3, 2, 1, 1), ('conv3', 128, 128, 3, 1, 1, 1),
                            ('conv34', 128, 256, 3, 2, 1, 1), ('conv4a', 256, 256, 3, 1, 1, 1), ('conv4b', 256, 256, 3, 1, 1, 1)])
        init_He(self)
        _imagenet_buffers(self)

    def forward(self, in_f, in_v):
        x = F.interpolate(torch.cat([(in_f - self.mean) / self.std, in_v], dim=1), size=(224, 224), mode='bilinear', align_corners=False)
        for name in self._order:
            x = getattr(self, name)(x)
        return x


class A_Regressor(nn.Module):
    """(feat1, feat2) -> theta (N,2,3); identity at initialisation (fc weight zero).  As in the reference, the second stride-2
    convolution is followed by conv5a / conv5b again: conv6a / conv6b hold parameters but are never applied
    (models/cpn_original.py:81-113)."""

    def __init__(self):
        super().__init__()
        _named_convs(self, [('conv45', 512, 512, 3, 2, 1, 1), ('conv5a', 512, 512, 3, 1, 1, 1), ('conv5b', 512, 512, 3, 1, 1, 1),
                            ('conv56', 512, 512, 3, 2, 1, 1), ('conv6a', 512, 512, 3, 1, 1, 1), ('conv6b', 512, 512, 3, 1, 1, 1)])
        init_He(self)
        self.fc = nn.Linear(512, 6)
        with torch.no_grad():
            self.fc.weight.zero_()
            self.fc.bias.copy_(torch.tensor([1., 0., 0., 0., 1., 0.]))

    def forward(self, feat1, feat2):
        x = torch.cat([feat1, feat2], dim=1)
        for name in ('conv45', 'conv5a', 'conv5b', 'conv56', 'conv5a', 'conv5b'):
            x = getattr(self, name)(x)
        return self.fc(F.avg_pool2d(x, x.shape[2]).flatten(1)).view(-1, 2, 3)


class Encoder(nn.Module):
    """Copy encoder, stride 4, 4 -> 128 channels.  models/cpn_original.py:117-136."""

    def __init__(self):
        super().__init__()
        _named_convs(self, [('conv12', 4, 64, 5, 2, 1, 1), ('conv2', 64, 64, 3, 1, 1, 1), ('conv23', 64, 128, 3, 2, 1, 1), ('conv3', 128, 128, 3, 1, 1, 1),
                            ('value3', 128, 128, 3, 1, 1, 0)])
        init_He(self)
        _imagenet_buffers(self)

    def forward(self, in_f, in_v):
        x = torch.cat([(in_f - self.mean) / self.std, in_v], dim=1)
        for name in ('conv12', 'conv2', 'conv23', 'conv3', 'value3'):
            x = getattr(self, name)(x)
        return x


class Decoder(nn.Module):
    """Paste decoder with the dilated block; output de-normalised.  models/cpn_original.py:140-183."""
    _low = ['conv4', 'conv5_1', 'conv5_2', 'convA4_1', 'convA4_2', 'convA4_3', 'convA4_4', 'conv3c', 'conv3b', 'conv3a']

    def __init__(self):
        super().__init__()
        _named_convs(self, [('conv4', 257, 257, 3, 1, 1, 1), ('conv5_1', 257, 257, 3, 1, 1, 1), ('conv5_2', 257, 257, 3, 1, 1, 1),
                            ('convA4_1', 257, 257, 3, 1, 2, 1), ('convA4_2', 257, 257, 3, 1, 4, 1), ('convA4_3', 257, 257, 3, 1, 8, 1), ('convA4_4', 257, 257, 3, 1, 16, 1),
                            ('conv3c', 257, 257, 3, 1, 1, 1), ('conv3b', 257, 128, 3, 1, 1, 1), ('conv3a', 128, 128, 3, 1, 1, 1),
                            ('conv32', 128, 64, 3, 1, 1, 1), ('conv2', 64, 64, 3, 1, 1, 1), ('conv21', 64, 3, 5, 1, 1, 0)])
        _imagenet_buffers(self)

    def forward(self, x):
        for name in self._low:
            x = getattr(self, name)(x)
        x = self.conv2(self.conv32(F.interpolate(x, scale_factor=2, mode='nearest')))
        x = self.conv21(F.interpolate(x, scale_factor=2, mode='nearest'))
        return x * self.std + self.mean


class CPNOriginal(nn.Module):
    """models/cpn_original.py:244-315 around the HIP affine warp and the HIP context matching.
    forward(x_target (B,3,H,W), m_target (B,1,H,W), x_refs (B,3,R,H,W), m_refs (B,1,R,H,W)) -> (y_hat, y_hat_comp)."""

    def __init__(self):
        super().__init__()
        self.A_Encoder = A_Encoder()
        self.A_Regressor = A_Regressor()
        self.Encoder = Encoder()
        self.CM_Module = CM_Module()
        self.Decoder = Decoder()

    def forward(self, x_target, m_target, x_refs, m_refs):
        x_refs_aligned, v_refs_aligned, _ = self.align(x_target, m_target, x_refs, m_refs)
        return self.copy_and_paste(x_target, 1 - m_target, x_refs_aligned, v_refs_aligned)

    def align(self, x_target, m_target, x_refs, m_refs):
        """-> (x_aligned (B,3,R,H,W), v_aligned (B,1,R,H,W) in {0,1}, v_maps).  models/cpn_original.py:258-290; theta ->
        positions inside the warp kernel, frames and visibility in one launch, v_maps by the fused kernel."""
        b, c, ref_n, h, w = x_refs.size()
        f_t = self.A_Encoder(x_target, m_target)
        f_r = self.A_Encoder(x_refs.transpose(1, 2).reshape(-1, c, h, w), m_refs.transpose(1, 2).reshape(-1, 1, h, w))
        f_t = f_t.unsqueeze(1).expand(b, ref_n, *f_t.shape[1:]).reshape(b * ref_n, *f_t.shape[1:])
        theta_rt = self.A_Regressor(f_t, f_r)   # target features first, as the reference has it
        if not torch.is_grad_enabled() and c == 3 and w >= 2:
            # inference (thesis_cpn/runner.py runs this under no_grad): frames, visibility (hole masks flipped per tap) and
            # v_maps (models/cpn_original.py:287) leave ONE launch
            return _ops.warp_infer(x_refs, m_refs, m_target, theta=theta_rt.float(), align_corners=False,
                                   v_mode=_lib.VMODE_BILINEAR_THRESHOLD)
        x_aligned, v_aligned, _ = _cpn.affine_align(x_refs, m_refs, None, theta_rt.float(), list(range(ref_n)))
        return x_aligned, v_aligned, _ops.vmap(v_aligned, 1 - m_target)

    def copy_and_paste(self, x_target, v_target, x_aligned, v_aligned):
        """-> (y_hat, y_hat_comp).  models/cpn_original.py:292-315.  (The encoder's mask channel of the target is
        `1 - v_target`, i.e. the hole mask, beside the references' visibility: kept as the reference has it.)"""
        b, c, f_ref, h, w = x_aligned.size()
        frames = torch.cat([x_target.unsqueeze(2), x_aligned], dim=2).transpose(1, 2).reshape(-1, c, h, w)
        masks = torch.cat([1 - v_target.unsqueeze(2), v_aligned], dim=2).transpose(1, 2).reshape(-1, 1, h, w)
        c_feats = self.Encoder(frames, masks)
        c_feats = c_feats.reshape(b, f_ref + 1, *c_feats.shape[1:]).transpose(1, 2)
        p_in, c_mask = self.CM_Module(c_feats.float(), v_target, v_aligned)
        y_hat = torch.clamp(self.Decoder(p_in), 0, 1)
        return y_hat, v_target * x_target + (1 - v_target) * y_hat
