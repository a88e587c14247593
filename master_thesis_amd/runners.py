"""The static step functions of the reference's runners on top of the HIP ops.

Mirrors `ThesisAlignmentRunner.train_step_propagate` (thesis_dfpn/runner.py:104-151) and `infer_step_propagate`
(thesis_dfpn/runner.py:153-159): same arguments, same returned tuple.  `model` is any callable with the alignment
network's interface, `model(x_target, m_target, x_refs, m_refs) -> (corr, flow_16, flow_64, flow_256)`
(models/thesis_alignment.py:177-207); the conv stacks behind it are not part of this package.  Everything after the model
call is the hot path: six `align_set` warps at 16 / 64 / 256 px, two nearest `resize_flow`s and the `v_map`s.
`resize_set` (utils/transforms.py:7-23) is caller-side glue on stock `F.interpolate`, kept as in the reference.
"""
import torch
import torch.nn.functional as F

from . import _ops
from .utils import flow as flow_ops


def resize_set(x, v, y, size):
    """(x, v, y) at size x size: bilinear for the frames, nearest for the visibility maps.  utils/transforms.py:7-23."""
    b, c, f, h, w = x.size()
    x_new = F.interpolate(x.transpose(1, 2).reshape(-1, c, h, w), (size, size), mode='bilinear').reshape(b, f, c, size, size).transpose(1, 2)
    v_new = F.interpolate(v.transpose(1, 2).reshape(-1, 1, h, w), (size, size)).reshape(b, f, 1, size, size).transpose(1, 2)
    y_new = F.interpolate(y.transpose(1, 2).reshape(-1, c, h, w), (size, size), mode='bilinear').reshape(b, f, c, size, size).transpose(1, 2)
    return x_new, v_new, y_new


def train_step_propagate(model, x, m, y, flow_gt, flows_use, t, r_list):
    """thesis_dfpn/runner.py:104-151.  x, y (B,3,F,H,W), m (B,1,F,H,W), flow_gt (B,F,H,W,2), flows_use (B,) bool."""
    corr, flow_16, flow_64, flow_256 = model(x[:, :, t], m[:, :, t], x[:, :, r_list], m[:, :, r_list])

    x_16, v_16, y_16 = resize_set(x, 1 - m, y, 16)
    x_64, v_64, y_64 = resize_set(x, 1 - m, y, 64)
    x_256, v_256, y_256 = x, 1 - m, y

    flow_256_gt = flow_gt[:, r_list]
    flow_16_gt = flow_ops.resize_flow(flow_256_gt, (16, 16))
    flow_64_gt = flow_ops.resize_flow(flow_256_gt, (64, 64))

    x_16_aligned_gt, v_16_aligned_gt = flow_ops.align_set(x_16[:, :, r_list], v_16[:, :, r_list], flow_16_gt)
    x_64_aligned_gt, v_64_aligned_gt = flow_ops.align_set(x_64[:, :, r_list], v_64[:, :, r_list], flow_64_gt)
    x_256_aligned_gt, v_256_aligned_gt = flow_ops.align_set(x_256[:, :, r_list], v_256[:, :, r_list], flow_256_gt)

    # target v_maps: only for the items whose ground-truth flow is usable, zero elsewhere
    use = flows_use.view(-1, 1, 1, 1, 1).to(v_64.dtype)
    v_map_64_gt = _ops.vmap(v_64_aligned_gt, v_64[:, :, t]) * use
    v_map_256_gt = _ops.vmap(v_256_aligned_gt, v_256[:, :, t]) * use

    x_16_aligned, v_16_aligned = flow_ops.align_set(x_16[:, :, r_list], v_16[:, :, r_list], flow_16)
    x_64_aligned, v_64_aligned = flow_ops.align_set(x_64[:, :, r_list], v_64[:, :, r_list], flow_64)
    x_256_aligned, v_256_aligned = flow_ops.align_set(x_256[:, :, r_list], v_256[:, :, r_list], flow_256)

    v_map_64 = _ops.vmap(v_64_aligned, v_64[:, :, t])
    v_map_256 = _ops.vmap(v_256_aligned, v_256[:, :, t])

    xs, vs, ys = (x_16, x_64, x_256), (v_16, v_64, v_256), (y_16, y_64, y_256)
    xs_aligned = (x_16_aligned, x_64_aligned, x_256_aligned)
    xs_aligned_gt = (x_16_aligned_gt, x_64_aligned_gt, x_256_aligned_gt)
    vs_aligned = (v_16_aligned, v_64_aligned, v_256_aligned)
    vs_aligned_gt = (v_16_aligned_gt, v_64_aligned_gt, v_256_aligned_gt)
    flows, flows_gt = (flow_16, flow_64, flow_256), (flow_16_gt, flow_64_gt, flow_256_gt)
    v_maps, v_maps_gt = (v_map_64, v_map_256), (v_map_64_gt, v_map_256_gt)
    return corr, xs, vs, ys, xs_aligned, xs_aligned_gt, vs_aligned, vs_aligned_gt, flows, flows_gt, flows_use, v_maps, v_maps_gt


def infer_step_propagate(model, x_target, m_target, x_ref, m_ref):
    """thesis_dfpn/runner.py:153-159: the unit the FF / IP / CP sequence schedulers call once per (target, reference) pair."""
    with torch.no_grad():
        *_, flow_256 = model(x_target, m_target, x_ref, m_ref)
    if x_ref.is_cuda and x_ref.size(1) == 3 and x_ref.size(-1) >= 2:
        # align_set + both `1 - m` + v_map in ONE launch (tcl_warp_fwd_fused): the schedulers call this unit thousands of times
        # per sequence at batch 1, where every extra kernel and tensor is host time
        return _ops.align_infer(x_ref, m_ref, flow_256, m_target)
    x_ref_aligned, v_ref_aligned = flow_ops.align_set(x_ref, 1 - m_ref, flow_256)
    v_map = _ops.vmap(v_ref_aligned, 1 - m_target)
    return x_ref_aligned, v_ref_aligned, v_map


def compute_loss(model_vgg, utils_losses, corr, xs, vs, ys, xs_aligned, flows, flows_gt, flows_use, t, r_list):
    """thesis_dfpn/runner.py:162-206, same arguments and return value `(total_loss, [corr_loss, flow_loss_16, flow_loss_64,
    flow_loss_256, alignment_recons_64, alignment_recons_256])`.

    The second call site of the cost volume (thesis_dfpn/runner.py:175): the volume of the ground-truth frames' features,
    masks None, under no_grad, is the target of an L1 term on the predicted volume.  `model_vgg(images) -> list` whose [3] is
    the (N,512,16,16) level; `utils_losses.masked_l1` as in utils/losses.py:26-32."""
    from .utils import correlation
    b, c, f, h, w = ys[2].size()
    n_ref = len(r_list)
    with torch.no_grad():
        y_in = ys[2].transpose(1, 2).reshape(b * f, c, h, w)
        if (h, w) != (256, 256):
            y_in = F.interpolate(y_in, (256, 256), mode='bilinear')
        y_feats = model_vgg(y_in)[3].float().reshape(b, f, -1, 16, 16).transpose(1, 2)
        corr_y = correlation.compute_masked_4d_correlation(y_feats[:, :, t], None, y_feats[:, :, r_list], None)
    corr_loss = F.l1_loss(corr.float(), corr_y)

    flow_losses = [utils_losses.masked_l1(flows[k].float(), flows_gt[k], torch.ones_like(flows_gt[k]), flows_use) for k in range(3)]

    recons = []
    for k in (1, 2):   # 64 and 256 px: target frame against each aligned reference, inside the target's visible area and the frame
        outside = ((flows[k] < -1).float() + (flows[k] > 1).float()).sum(4).clamp(0, 1).unsqueeze(1)
        target = xs[k][:, :, t].unsqueeze(2).expand(-1, -1, n_ref, -1, -1)
        visible = vs[k][:, :, t].unsqueeze(2).expand(-1, -1, n_ref, -1, -1) * (1 - outside)
        recons.append(utils_losses.masked_l1(target, xs_aligned[k], visible, reduction='sum'))

    items = [corr_loss] + flow_losses + recons
    total_loss = corr_loss + flow_losses[0] + flow_losses[1] + flow_losses[2]
    total_loss = total_loss + recons[0] + recons[1]
    return total_loss, items


class ThesisAlignmentRunner:
    """Namespace with the reference class's static step functions (thesis_dfpn/runner.py:104-206)."""
    train_step_propagate = staticmethod(train_step_propagate)
    infer_step_propagate = staticmethod(infer_step_propagate)
    compute_loss = staticmethod(compute_loss)


class ThesisCPNRunner:
    @staticmethod
    def infer_alignment_step_propagate(model, x_target, m_target, x_refs, m_refs):
        """thesis_cpn/runner.py:96-98: the CPN aligner's own `align` (affine warp; models/cpn_original.py:260-290)."""
        return model.align(x_target, m_target, x_refs, m_refs)


def _align_step(model_alignment, x_target, m_target, x_ref, m_ref):
    # the reference dispatches on isinstance(ThesisAlignmentModel) (thesis_inpainting/runner.py:88-95); the dense-flow
    # aligner is a plain callable here, the CPN aligner is recognised by its `align` method
    if hasattr(model_alignment, 'align'):
        return ThesisCPNRunner.infer_alignment_step_propagate(model_alignment, x_target, m_target, x_ref, m_ref)
    return infer_step_propagate(model_alignment, x_target, m_target, x_ref, m_ref)


class ThesisInpaintingRunner:
    """Static step functions of thesis_inpainting/runner.py:85-128.  `model_alignment` is frozen (no_grad); `model` is the
    inpainting network `model(x_target, v_target, x_ref_aligned, v_ref_aligned, v_map) -> (y_hat, y_hat_comp)`
    (models/thesis_inpainting.py:14-36)."""

    @staticmethod
    def train_step_propagate(model_alignment, model, x_target, m_target, x_ref, m_ref):
        with torch.no_grad():
            x_ref_aligned, v_ref_aligned, v_map = _align_step(model_alignment, x_target, m_target, x_ref, m_ref)
        y_hat, y_hat_comp = model(x_target, 1 - m_target, x_ref_aligned, v_ref_aligned, v_map)
        return x_ref_aligned, v_ref_aligned, v_map, y_hat, y_hat_comp

    @staticmethod
    def infer_step_propagate(model_alignment, model, x_target, m_target, x_ref, m_ref):
        with torch.no_grad():
            return ThesisInpaintingRunner.train_step_propagate(model_alignment, model, x_target, m_target, x_ref, m_ref)

    @staticmethod
    def compute_loss(utils_losses, y_target, v_target, y_hat, y_hat_comp, v_map):
        """Non-hole / visible-hole / non-visible-hole L1 + perceptual + gradient terms, weights 0.5 / 2 / 1 / 0.5 / 1.
        `utils_losses` provides masked_l1, perceptual and grad (utils/losses.py); thesis_inpainting/runner.py:113-128."""
        b, c, h, w = y_target.size()
        f = y_hat.size(2)
        target_img = y_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
        nh_mask = v_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
        vh_mask = v_map
        nvh_mask = (1 - nh_mask) - vh_mask
        loss_nh = utils_losses.masked_l1(y_hat, target_img, nh_mask, reduction='sum', weight=0.50)
        loss_vh = utils_losses.masked_l1(y_hat, target_img, vh_mask, reduction='sum', weight=2)
        loss_nvh = utils_losses.masked_l1(y_hat_comp, target_img, nvh_mask, reduction='sum', weight=1)
        loss_perceptual, *_ = utils_losses.perceptual(
            y_hat.transpose(1, 2).reshape(-1, c, h, w), target_img.transpose(1, 2).reshape(-1, c, h, w), weight=0.50)
        loss_grad = utils_losses.grad(y_hat.squeeze(2), target_img.squeeze(2), reduction='mean', weight=1)
        loss = loss_nh + loss_vh + loss_nvh + loss_perceptual + loss_grad
        return loss, [loss_nh, loss_vh, loss_nvh, loss_perceptual, loss_grad]
