"""The loss terms the runners' `compute_loss` functions call (`LossesUtils` of the reference's utils/losses.py:7-80: same
method names, arguments, weights and normalisations), written on stock torch ops.  Caller-side glue, not the hot path.
`model_vgg` is any callable returning a list of feature maps (the pretrained VGG-16 of models/vgg_16.py needs torchvision
weights, which do not ship here)."""
import torch
import torch.nn.functional as F

_SOBEL = ((1., 0., -1.), (2., 0., -2.), (1., 0., -1.))   # d/dx; its transpose is d/dy


def _pick(batch_mask, *tensors):
    """Rows of the batch selected by a boolean list / tensor (None = all); None when nothing is selected."""
    if batch_mask is None:
        return tensors
    if not any(batch_mask):
        return None
    return tuple(t[batch_mask] for t in tensors)


def _normalised_gram(feat):
    b, c, h, w = feat.size()
    flat = feat.reshape(b * c, h * w)
    return torch.mm(flat, flat.t()) / (b * c * h * w)


class LossesUtils:

    def __init__(self, model_vgg, device):
        self.model_vgg = model_vgg
        kx = torch.tensor(_SOBEL, dtype=torch.float32, device=device)
        # one depthwise filter pair per RGB channel, as the reference's grad_horz / grad_vert (3,1,3,3) banks
        self.grad_horz = kx.expand(3, 1, 3, 3).contiguous()
        self.grad_vert = kx.t().expand(3, 1, 3, 3).contiguous()

    # -- pixel losses
    def masked_l1(self, input, target, mask, batch_mask=None, reduction='mean', weight=1):
        """L1 between the masked tensors; reduction='sum' is normalised by the mask area (+1e-9).  An all-False batch_mask
        gives a zero of shape (1,).  utils/losses.py:26-32"""
        picked = _pick(batch_mask, input, target, mask)
        if picked is None:
            return torch.zeros(1).to(input.device)
        x, y, k = picked
        value = F.l1_loss(x * k, y * k, reduction=reduction)
        if reduction == 'sum':
            value = value / (torch.sum(k) + 1E-9)
        return weight * value

    def bce(self, input, target, mask, batch_mask=None, reduction='mean', weight=1):
        """utils/losses.py:34-38 (no epsilon on the area here, and no early exit for an empty selection)"""
        x, y, k = (input, target, mask) if batch_mask is None else (input[batch_mask], target[batch_mask], mask[batch_mask])
        value = F.binary_cross_entropy(x * k, y * k, reduction=reduction)
        if reduction == 'sum':
            value = value / torch.sum(k)
        return weight * value

    # -- feature losses
    def perceptual(self, input, target, weight=1):
        """mean over the feature levels of the L1 distance -> (loss, input features, target features).  utils/losses.py:40-46"""
        feats_in, feats_tg = self.model_vgg(input.contiguous()), self.model_vgg(target.contiguous())
        per_level = [F.l1_loss(a, b) for a, b in zip(feats_in, feats_tg)]
        return sum(per_level) * weight / len(feats_in), feats_in, feats_tg

    def style(self, x_vgg, x_hat_vgg, weight=1):
        """L1 between the normalised Gram matrices (over batch x channel rows, as the reference has it).  utils/losses.py:48-55"""
        per_level = [F.l1_loss(_normalised_gram(b), _normalised_gram(a)) for a, b in zip(x_vgg, x_hat_vgg)]
        return sum(per_level) * weight / len(x_vgg)

    # -- smoothness
    def tv(self, x_hat, weight=1):
        """squared forward differences along h and w over the element count.  utils/losses.py:57-61"""
        energy = torch.diff(x_hat, dim=2).pow(2).sum() + torch.diff(x_hat, dim=3).pow(2).sum()
        return energy / x_hat.numel() * weight

    def grad(self, input, target, reduction, weight=1):
        """masked_l1 (all-ones mask) between the Sobel responses [d/dx of RGB, d/dy of RGB].  utils/losses.py:63-67"""
        sobel = lambda t: torch.cat((self._grad_horizontal(t), self._grad_vertical(t)), dim=1)
        g_in, g_tg = sobel(input), sobel(target)
        return self.masked_l1(g_in, g_tg, torch.ones_like(g_in), reduction=reduction, weight=weight)

    def _grad_horizontal(self, x):
        return F.conv2d(x, self.grad_horz, padding=1, groups=3)

    def _grad_vertical(self, x):
        return F.conv2d(x, self.grad_vert, padding=1, groups=3)
