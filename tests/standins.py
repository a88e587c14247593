"""Deterministic stand-ins for the networks around the hot path, in a torch and a numpy flavour.

The sequence schedulers and runner step functions take the alignment network and the inpainting network as arguments.
To pin OUR schedulers against the reference's (tests/golden/make_golden_sequences.py runs the real reference loops) both
sides need the same "networks": closed-form functions with no weights.  The torch flavour is device-agnostic (CPU inside
the golden generator, GPU in the parity tests); the numpy flavour feeds the oracle.

  aligner      : flow_256 = identity grid + a constant shift of (+1.3, -0.7) pixels  (far from the .5 rounding ties
                 of the nearest-neighbour visibility warp, so masks stay bit-exact across float implementations)
  cpn aligner  : `.align` with a constant theta (small rotation + shift), through the affine warp
  inpainter    : y_hat = clamp(0.2 + 0.5 * x_ref_aligned * v_ref_aligned + 0.3 * x_target, 0, 1);  y_hat_comp as
                 models/thesis_inpainting.py:33
  cpn model    : (y_hat, y_hat_comp) with y_hat = mean over refs of x_refs * (1 - m_refs)
"""
import numpy as np

SHIFT_PX = (1.3, -0.7)
THETA = ((0.98, 0.05, 0.06), (-0.04, 1.01, -0.03))


# ------------------------------------------------------------------------------------------------ torch flavour
def torch_identity_grid(h, w, device):
    import torch
    ys = torch.linspace(-1, 1, h, device=device).view(h, 1).expand(h, w)
    xs = torch.linspace(-1, 1, w, device=device).view(1, w).expand(h, w)
    return torch.stack([xs, ys], dim=2)


def torch_flow(x_target, m_target, x_ref, m_ref):
    """Call signature of ThesisAlignmentModel.forward (models/thesis_alignment.py:177-207) -> (corr, f16, f64, f256)."""
    import torch
    b, c, f, h, w = x_ref.size()
    shift = torch.tensor([SHIFT_PX[0] * 2 / (w - 1), SHIFT_PX[1] * 2 / (h - 1)], device=x_ref.device)
    flow = (torch_identity_grid(h, w, x_ref.device) + shift).view(1, 1, h, w, 2).repeat(b, f, 1, 1, 1)
    return None, None, None, flow


def torch_inpaint(x_target, v_target, x_refs_aligned, v_refs_aligned, v_maps):
    """Call signature of ThesisInpaintingVisible.forward (models/thesis_inpainting.py:14-36)."""
    import torch
    f = x_refs_aligned.size(2)
    x_t = x_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
    v_t = v_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
    y_hat = torch.clamp(0.2 + 0.5 * x_refs_aligned * v_refs_aligned + 0.3 * x_t, 0, 1)
    return y_hat, v_t * x_t + (1 - v_t) * y_hat


def torch_cpn_model(x_target, m_target, x_refs, m_refs):
    """Call signature of CPNOriginal.forward (models/cpn_original.py:255-258)."""
    y_hat = (x_refs * (1 - m_refs)).mean(2)
    return y_hat, (1 - m_target) * x_target + m_target * y_hat


def torch_theta(n, device):
    import torch
    return torch.tensor(THETA, device=device).view(1, 2, 3).repeat(n, 1, 1)


# ------------------------------------------------------------------------------------------------ numpy flavour
def np_flow(x_ref):
    b, c, f, h, w = x_ref.shape
    xs = np.linspace(-1, 1, w, dtype=np.float64).astype(np.float32)
    ys = np.linspace(-1, 1, h, dtype=np.float64).astype(np.float32)
    grid = np.stack([np.broadcast_to(xs[None, :], (h, w)), np.broadcast_to(ys[:, None], (h, w))], axis=2).astype(np.float32)
    shift = np.array([np.float32(SHIFT_PX[0] * 2 / (w - 1)), np.float32(SHIFT_PX[1] * 2 / (h - 1))], dtype=np.float32)
    return np.broadcast_to((grid + shift)[None, None], (b, f, h, w, 2)).astype(np.float32).copy()


def np_inpaint(x_target, v_target, x_refs_aligned, v_refs_aligned, v_maps):
    f = x_refs_aligned.shape[2]
    x_t = np.repeat(x_target[:, :, None], f, axis=2)
    v_t = np.repeat(v_target[:, :, None], f, axis=2)
    y_hat = np.clip(np.float32(0.2) + np.float32(0.5) * x_refs_aligned * v_refs_aligned + np.float32(0.3) * x_t, 0, 1).astype(np.float32)
    return y_hat, (v_t * x_t + (1 - v_t) * y_hat).astype(np.float32)


def np_cpn_model(x_target, m_target, x_refs, m_refs):
    y_hat = (x_refs * (1 - m_refs)).mean(2, dtype=np.float32)
    return y_hat, ((1 - m_target) * x_target + m_target * y_hat).astype(np.float32)


def np_theta(n):
    return np.broadcast_to(np.array(THETA, dtype=np.float32)[None], (n, 2, 3)).copy()


# ------------------------------------------------------------------------------------------------ inputs
def moving_hole_sequence(seed, c, f, h, w, hole, step):
    """x (c,f,h,w) U[0,1) and m (1,f,h,w): a `hole` x `hole` square drifting by `step` px per frame (plus a second,
    static one on odd seeds so that some pixels are visible in no neighbour)."""
    rng = np.random.RandomState(seed)
    x = rng.rand(c, f, h, w).astype(np.float32)
    m = np.zeros((1, f, h, w), np.float32)
    y0, x0 = rng.randint(1, h - hole - 1), rng.randint(1, max(2, w // 3))
    for t in range(f):
        xx = min(max(x0 + step * t, 0), w - hole)
        m[0, t, y0:y0 + hole, xx:xx + hole] = 1
    if seed % 2 == 1:
        m[0, :, h - 4:h - 1, w - 4:w - 1] = 1
    return x, m


# ---- CPNet golden (tests/golden/make_golden_cpnet.py and tests/test_cpnet.py share these)
CPNET_SEED = 11


def cpnet_package_net():
    """This package's CPNet('full') with seeded weights: theta away from the identity (shifts of a few pixels), and a
    variance-preserving init of the encoder / decoder (with torch's default one y_hat is set by the biases alone)."""
    import torch
    from master_thesis_amd.models import cpn as my_cpn
    torch.manual_seed(CPNET_SEED)
    net = my_cpn.CPNet('full')
    with torch.no_grad():
        net.alignment_regressor.fc.weight.mul_(0.5)
        for mod in list(net.encoder.modules()) + list(net.decoder.modules()):
            if isinstance(mod, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(mod.weight, nonlinearity='relu')
    return net.eval()


def cpnet_inputs():
    import torch
    g = torch.Generator().manual_seed(CPNET_SEED + 1)
    B, F_, H, W = 1, 3, 64, 64
    x = torch.rand(B, 3, F_, H, W, generator=g)
    y = torch.rand(B, 3, F_, H, W, generator=g)
    m = (torch.rand(B, 1, F_, H // 8, W // 8, generator=g) < 0.25).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    return x * (1 - m), m, y, 1, [0, 2]


# ---- ThesisAlignmentModel golden (tests/golden/make_golden_alignment_model.py and tests/test_alignment_model.py)
ALIGN_SEED = 23


def alignment_package_net():
    """This package's ThesisAlignmentModel around the seeded stand-in feature extractor, variance-preserving init
    for the conv stacks (with torch's default one the flows are set by the biases alone)."""
    import torch
    from master_thesis_amd import harness_c3
    from master_thesis_amd.models import thesis_alignment as ta
    torch.manual_seed(ALIGN_SEED)
    net = ta.ThesisAlignmentModel(harness_c3.FeatureStandIn())
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                torch.nn.init.kaiming_normal_(mod.weight, nonlinearity='relu')
        for stack in (net.corr_mixer.mixer, net.flow_64.nn, net.flow_256.nn):   # flows of moderate size
            stack[-1].weight.mul_(0.1)
    return net.eval()


def alignment_inputs():
    import torch
    g = torch.Generator().manual_seed(ALIGN_SEED + 1)
    B, R, H, W = 1, 2, 64, 96
    x_target = torch.rand(B, 3, H, W, generator=g)
    x_refs = torch.rand(B, 3, R, H, W, generator=g)
    m_target = (torch.rand(B, 1, H // 16, W // 16, generator=g) < 0.2).float().repeat_interleave(16, 2).repeat_interleave(16, 3)
    m_refs = (torch.rand(B, 1, R, H // 16, W // 16, generator=g) < 0.2).float().repeat_interleave(16, 3).repeat_interleave(16, 4)
    return x_target * (1 - m_target), m_target, x_refs * (1 - m_refs), m_refs


# ---- CPNOriginal golden (tests/golden/make_golden_cpn_original.py and tests/test_cpn_original.py)
CPNO_SEED = 31


def cpn_original_package_net():
    """This package's CPNOriginal with seeded weights; the regressor's last layer (zero in a fresh net = identity theta) is
    perturbed so that the warp does something, the decoder gets a variance-preserving init."""
    import torch
    from master_thesis_amd.models import cpn_original as my
    torch.manual_seed(CPNO_SEED)
    net = my.CPNOriginal()
    with torch.no_grad():
        net.A_Regressor.fc.weight.normal_(0, 0.01)
        for mod in net.Decoder.modules():
            if isinstance(mod, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(mod.weight, nonlinearity='relu')
    return net.eval()


def cpn_original_inputs():
    import torch
    g = torch.Generator().manual_seed(CPNO_SEED + 1)
    B, R, H, W = 2, 3, 48, 64
    x_target = torch.rand(B, 3, H, W, generator=g)
    x_refs = torch.rand(B, 3, R, H, W, generator=g)
    m_target = (torch.rand(B, 1, H // 8, W // 8, generator=g) < 0.25).float().repeat_interleave(8, 2).repeat_interleave(8, 3)
    m_refs = (torch.rand(B, 1, R, H // 8, W // 8, generator=g) < 0.25).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    return x_target * (1 - m_target), m_target, x_refs * (1 - m_refs), m_refs
