"""Drop-in for the reference's utils/losses.py `LossesUtils`: the loss terms the runners' `compute_loss` functions call.

Caller-side glue, not the hot path: reductions and 3x3 Sobel filters on stock torch ops, same names, arguments and
weights as utils/losses.py:7-80.  `model_vgg` is any callable returning a list of feature maps (the pretrained VGG-16 of
models/vgg_16.py needs torchvision weights, which do not ship here)."""
import torch
import torch.nn.functional as F


class LossesUtils:

    def __init__(self, model_vgg, device):
        self.model_vgg = model_vgg
        sobel = torch.tensor([[1, 0, -1], [2, 0, -2], [1, 0, -1]], dtype=torch.float32)
        self.grad_horz = sobel.view(1, 1, 3, 3).repeat(3, 1, 1, 1).to(device)
        self.grad_vert = sobel.t().contiguous().view(1, 1, 3, 3).repeat(3, 1, 1, 1).to(device)

    def masked_l1(self, input, target, mask, batch_mask=None, reduction='mean', weight=1):
        """L1 over mask == 1; with reduction='sum' normalised by the mask area.  utils/losses.py:26-32"""
        if batch_mask is not None:
            if not any(batch_mask):
                return torch.zeros(1).to(input.device)
            input, target, mask = input[batch_mask], target[batch_mask], mask[batch_mask]
        loss = F.l1_loss(input * mask, target * mask, reduction=reduction)
        return weight * loss / (torch.sum(mask) + 1E-9 if reduction == 'sum' else 1)

    def bce(self, input, target, mask, batch_mask=None, reduction='mean', weight=1):
        if batch_mask is not None:
            input, target, mask = input[batch_mask], target[batch_mask], mask[batch_mask]
        loss = F.binary_cross_entropy(input * mask, target * mask, reduction=reduction)
        return weight * loss / (torch.sum(mask) if reduction == 'sum' else 1)

    def perceptual(self, input, target, weight=1):
        input_vgg, target_vgg = self.model_vgg(input.contiguous()), self.model_vgg(target.contiguous())
        loss = sum(F.l1_loss(a, b) for a, b in zip(input_vgg, target_vgg))
        return loss * weight / len(input_vgg), input_vgg, target_vgg

    def style(self, x_vgg, x_hat_vgg, weight=1):
        loss = 0
        for a, a_hat in zip(x_vgg, x_hat_vgg):
            b, c, h, w = a.size()
            gram = torch.mm(a.reshape(b * c, h * w), a.reshape(b * c, h * w).t())
            gram_hat = torch.mm(a_hat.reshape(b * c, h * w), a_hat.reshape(b * c, h * w).t())
            loss = loss + F.l1_loss(gram_hat / (b * c * h * w), gram / (b * c * h * w))
        return loss * weight / len(x_vgg)

    def tv(self, x_hat, weight=1):
        dh = (x_hat[:, :, 1:, :] - x_hat[:, :, :-1, :]).pow(2).sum()
        dw = (x_hat[:, :, :, 1:] - x_hat[:, :, :, :-1]).pow(2).sum()
        return (dh + dw) / (x_hat.size(0) * x_hat.size(1) * x_hat.size(2) * x_hat.size(3)) * weight

    def grad(self, input, target, reduction, weight=1):
        gi = torch.cat((self._grad_horizontal(input), self._grad_vertical(input)), dim=1)
        gt = torch.cat((self._grad_horizontal(target), self._grad_vertical(target)), dim=1)
        return self.masked_l1(gi, gt, torch.ones_like(gi), batch_mask=None, reduction=reduction, weight=weight)

    def _grad_horizontal(self, x):
        return F.conv2d(x, padding=1, weight=self.grad_horz, groups=3)

    def _grad_vertical(self, x):
        return F.conv2d(x, padding=1, weight=self.grad_vert, groups=3)
