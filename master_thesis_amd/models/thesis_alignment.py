"""Drop-in for the reference's models/thesis_alignment.py (the dense-flow alignment network, DFPN): same class names,
constructor arguments, sub-module names and state_dict keys, so a checkpoint of the reference loads unchanged.  The cost
volume is `utils.correlation.compute_masked_4d_correlation` on the HIP kernels; the convolution stacks are stock torch
convolutions (they are not on the hand-written path).  Pinned against the reference's own forward by
tests/golden/alignment_model.npz."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils import correlation


def _stack(spec, in_c):
    """spec rows: (out_c, kernel, stride, relu) | ('T', out_c, kernel, relu) for a stride-2 transposed convolution."""
    layers, c = [], in_c
    for row in spec:
        if row[0] == 'T':
            _, out_c, k, relu = row
            layers.append(nn.ConvTranspose2d(c, out_c, k, stride=2, padding=k // 2, output_padding=1))
        else:
            out_c, k, stride, relu = row
            layers.append(nn.Conv2d(c, out_c, k, stride=stride, padding=k // 2))
        if relu:
            layers.append(nn.ReLU())
        c = out_c
    return nn.Sequential(*layers)


class SeparableConv4d(nn.Module):
    """Two 2-D stacks over a (b,t,h,w,h,w) volume: over the reference-pixel plane of every target pixel, then over the
    target-pixel plane of every reference pixel.  models/thesis_alignment.py:8-36."""

    def __init__(self, in_c=1, out_c=1):
        super().__init__()
        self.conv_1 = _stack([(128, 3, 1, 1), (256, 3, 1, 1), (256, 3, 1, 0)], in_c)
        self.conv_2 = _stack([(256, 3, 1, 1), (128, 3, 1, 1), (out_c, 3, 1, 0)], 256)

    def forward(self, corr):
        b, t, h, w = corr.shape[:4]
        x = self.conv_1(corr.reshape(b * t * h * w, -1, h, w))
        x = x.reshape(b, t, h * w, x.size(1), h * w).permute(0, 1, 4, 3, 2)
        x = self.conv_2(x.reshape(b * t * h * w, x.size(3), h, w))
        return x.float().reshape(b, t, h, w, h, w).permute(0, 1, 4, 5, 2, 3)   # (.float(): a no-op unless the stacks ran under bf16 autocast)


class Softmax3d(nn.Module):
    """Softmax over (reference h, reference w, frame) jointly, per target pixel.  models/thesis_alignment.py:39-57 (A15)."""

    def forward(self, input):
        assert input.dim() == 6   # (B,T,H1,W1,H,W)
        b, t, h, w = input.shape[:4]
        p = F.softmax(input.permute(0, 2, 3, 4, 5, 1).reshape(b, h, w, -1), dim=3)
        return p.reshape(b, h, w, h, w, t).permute(0, 5, 1, 2, 3, 4)


class CorrelationVGG(nn.Module):
    """Frozen features -> masked cost volume (HIP) -> SeparableConv4d.  models/thesis_alignment.py:60-93."""

    def __init__(self, model_vgg, use_softmax=False):
        super().__init__()
        self.model_vgg = model_vgg
        self.conv = SeparableConv4d()
        self.softmax = Softmax3d() if use_softmax else None

    def forward(self, x_target, m_target, x_ref, m_ref):
        b, c, ref_n, h, w = x_ref.size()
        with torch.no_grad():
            f_t = self.model_vgg(x_target, normalize_input=False)[3]
            f_r = self.model_vgg(x_ref.transpose(1, 2).reshape(b * ref_n, c, h, w), normalize_input=False)[3]
        f_r = f_r.reshape(b, ref_n, -1, 16, 16).transpose(1, 2)
        fh, fw = f_r.shape[3:]
        v_t = F.interpolate(1 - m_target, size=(fh, fw), mode='nearest')
        v_r = F.interpolate(1 - m_ref.transpose(1, 2).reshape(b * ref_n, 1, h, w), size=(fh, fw), mode='nearest') \
            .reshape(b, ref_n, 1, fh, fw).transpose(1, 2)
        corr = self.conv(correlation.compute_masked_4d_correlation(f_t.float(), v_t.float(), f_r.float(), v_r.float()))
        return self.softmax(corr) if self.softmax else corr


class AlignmentCorrelationMixer(nn.Module):
    """(b,f,16,16,16,16) volume -> 16 x 16 flow (b,f,16,16,2).  models/thesis_alignment.py:96-120."""

    def __init__(self, corr_size=16):
        super().__init__()
        assert corr_size == 16
        s = corr_size
        spec = []
        for c_in_out in ((s * s, s), (s, s // 2), (s // 2, s // 4)):
            spec += [(c_in_out[0], 5, 1, 1), (c_in_out[0], 3, 1, 1), (c_in_out[1], 3, 1, 1)]
        spec += [(s // 4, 5, 1, 1), (s // 4, 3, 1, 1), (s // 8, 3, 1, 0), (s // 8, 5, 1, 0), (s // 8, 3, 1, 0)]
        self.mixer = _stack(spec, s * s)

    def forward(self, corr):
        b, f, h, w = corr.shape[:4]
        return self.mixer(corr.reshape(b * f, -1, 16, 16)).float().reshape(b, f, 2, h, w).permute(0, 1, 3, 4, 2)


class FlowEstimator(nn.Module):
    """[reference frame, target frame, both masks, previous flow] (10 channels) -> flow at the same size.
    models/thesis_alignment.py:123-163."""

    def __init__(self, in_c=10):
        super().__init__()
        down = lambda k: [(128, 5, 1, 1), (128, 3, 1, 1), (128, k, 2, 1)]
        up = lambda k: [(128, 5, 1, 1), (128, 3, 1, 1), ('T', 128, k, 1)]
        self.nn = _stack(down(5) + down(3) + down(3) + up(3) + up(3) + up(5) + [(128, 5, 1, 1), (128, 3, 1, 1), (2, 3, 1, 0)], in_c)

    def forward(self, x_target, m_target, x_ref, m_ref, flow_pre):
        b, c, ref_n, h, w = x_ref.size()
        per = lambda t: t.transpose(1, 2).reshape(b * ref_n, t.size(1), h, w)
        rep = lambda t: t.unsqueeze(1).expand(b, ref_n, t.size(1), h, w).reshape(b * ref_n, t.size(1), h, w)
        x = torch.cat([per(x_ref), rep(x_target), per(m_ref), rep(m_target), flow_pre.reshape(b * ref_n, h, w, 2).permute(0, 3, 1, 2)], dim=1)
        return self.nn(x).float().reshape(b, ref_n, 2, h, w).permute(0, 1, 3, 4, 2)   # flows leave an autocast region as fp32


class ThesisAlignmentModel(nn.Module):
    """model(x_target (B,3,H,W), m_target (B,1,H,W), x_refs (B,3,R,H,W), m_refs (B,1,R,H,W)) ->
    (corr (B,R,16,16,16,16), flow_16, flow_64, flow at (H,W)).  models/thesis_alignment.py:166-225."""

    def __init__(self, model_vgg):
        super().__init__()
        self.corr = CorrelationVGG(model_vgg)
        self.corr_mixer = AlignmentCorrelationMixer()
        self.flow_64 = FlowEstimator()
        self.flow_256 = FlowEstimator()
        self.register_buffer('mean', torch.as_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1))
        self.register_buffer('std', torch.as_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1))

    def forward(self, x_target, m_target, x_refs, m_refs):
        b, c, ref_n, h, w = x_refs.size()
        x_target = (x_target - self.mean.squeeze(2)) / self.std.squeeze(2)
        x_refs = (x_refs - self.mean) / self.std
        sq = self.interpolate_data(x_target, m_target, x_refs, m_refs, 256, 256)
        corr = self.corr(*sq)
        flow_16 = self.corr_mixer(corr)
        flow_64 = self.flow_64(*self.interpolate_data(x_target, m_target, x_refs, m_refs, 64, 64), self.interpolate_flow(flow_16, 64, 64))
        flow_256 = self.flow_256(*sq, self.interpolate_flow(flow_64, 256, 256))
        return corr, flow_16, flow_64, self.interpolate_flow(flow_256, h, w)

    def interpolate_data(self, x_target, m_target, x_ref, m_ref, h_new, w_new):
        b, c, ref_n, h, w = x_ref.size()
        if (h, w) == (h_new, w_new):
            return x_target, m_target, x_ref, m_ref
        frames = lambda t, mode: F.interpolate(t.transpose(1, 2).reshape(b * ref_n, t.size(1), h, w), (h_new, w_new), mode=mode) \
            .reshape(b, ref_n, t.size(1), h_new, w_new).transpose(1, 2)
        return (F.interpolate(x_target, (h_new, w_new), mode='bilinear'), F.interpolate(m_target, (h_new, w_new), mode='nearest'),
                frames(x_ref, 'bilinear'), frames(m_ref, 'nearest'))

    def interpolate_flow(self, flow, h_new, w_new):
        b, ref_n, h, w, _ = flow.size()
        return F.interpolate(flow.reshape(b * ref_n, h, w, 2).permute(0, 3, 1, 2).float(), (h_new, w_new), mode='bilinear') \
            .reshape(b, ref_n, 2, h_new, w_new).permute(0, 1, 3, 4, 2)
