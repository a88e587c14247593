"""runner step, sequence schedulers, movement simulator.

CPU: the numpy oracle (oracle/sequence_oracle.py) and the product's pure-host schedule functions against golden vectors
captured from the real reference (tests/golden/make_golden_sequences.py).
GPU: the batched schedulers / step functions / simulator on the HIP ops against the same golden vectors and against the
oracle run one sequence at a time (different sequences stop at different slots, which is what the batching must get right).
"""
import numpy as np
import pytest
import torch

from oracle import sequence_oracle as SO
from tests import standins
from tests.conftest import assert_close, assert_exact, load_golden

gpu = pytest.mark.gpu


def _kw(case):
    return {str(k): int(v) for k, v in zip(case["kw_keys"], case["kw_vals"])}


def _np_step(aligner):
    if aligner == "flow":
        align = lambda xt, mt, xr, mr: SO.dfpn_infer_step(standins.np_flow, xt, mt, xr, mr)
    else:
        align = lambda xt, mt, xr, mr: SO.cpn_align_step(standins.np_theta, xt, mt, xr, mr)
    return lambda xt, mt, xr, mr: SO.inpaint_infer_step(align, standins.np_inpaint, xt, mt, xr, mr)


def _oracle_sequence(alg, aligner, x, m, kw):
    if alg == "cpn":
        return SO.algorithm_cpn(x, m, standins.np_cpn_model)
    return getattr(SO, "algorithm_" + alg)(x, m, _np_step(aligner), **kw)


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("case", load_golden("schedules"))
def test_schedules_oracle_and_product(case):
    from master_thesis_amd import schedulers as S
    kind, args, ref = str(case["kind"]), [int(a) for a in case["args"]], [int(v) for v in case["out"]]
    if kind == "ff":
        assert SO.ff_indexes(*args) == ref
        assert S.inpainting_algorithm_ff_indexes(*args) == ref
    elif kind == "ip":
        t, max_t, s, D = args
        t_list = sorted(range(max_t), key=lambda k: abs(k - max_t // 2))
        assert SO.ip_indexes(t, t_list, s, D) == ref
        assert S.inpainting_algorithm_ip_indexes(t, t_list, s, D) == ref
    else:
        assert SO.cpn_reference_indexes(*args) == ref
        assert S.get_reference_frame_indexes(*args) == ref


@pytest.mark.parametrize("case", load_golden("sequences"))
def test_oracle_sequences_vs_reference(case):
    f, h, w, hole, step = (int(v) for v in case["dims"])
    x, m = standins.moving_hole_sequence(int(case["seed"]), 3, f, h, w, hole, step)
    y = _oracle_sequence(str(case["alg"]), str(case["aligner"]), x, m, _kw(case))
    assert_close(y, case["y"], what="oracle %s/%s" % (case["alg"], case["aligner"]))


@pytest.mark.parametrize("case", load_golden("inpaint_step"))
def test_oracle_inpaint_step_vs_reference(case):
    outs = _np_step(str(case["aligner"]))(case["x_t"], case["m_t"], case["x_r"], case["m_r"])
    for got, key in zip(outs, ["x_ref_aligned", "v_ref_aligned", "v_map", "y_hat", "y_hat_comp"]):
        (assert_exact if key.startswith("v_") else assert_close)(got, case[key], what="oracle step " + key)


@pytest.mark.parametrize("case", load_golden("movement"))
def test_oracle_movement_vs_reference(case):
    n = case["mats"].shape[0]
    h, w = case["x"].shape[1:]
    stacked = SO.stack_transformations(case["mats"], n // 2)
    assert_close(stacked, case["stacked"], rtol=1e-5, atol=1e-6, what="stack_transformations")
    assert_close(np.stack([SO.affine2theta(a, h, w) for a in case["stacked"]]), case["theta"], rtol=1e-6, atol=1e-7, what="affine2theta")
    y, flow_inv = SO.simulate_movement(case["x"], case["mats"])
    assert_close(y, case["y"], what="simulate_movement y")
    assert_close(flow_inv, case["flow_inv"], what="simulate_movement flow_inv")
    assert_close(SO.transform_single(case["x"], case["mats"][0]), case["single"], what="transform_single")


def test_product_movement_host_chain():
    """The 3x3 chain and theta conversion run on the host with the reference's float32 torch ops.  The elementwise theta
    conversion is bit-identical; the inverse / matmul chain goes through the host's LAPACK / BLAS and may differ in the
    last bit between CPU models (seen: 1 ulp between the build container and the GPU box)."""
    from master_thesis_amd.utils.movement import MovementSimulator, affines2theta
    for case in load_golden("movement"):
        n = case["mats"].shape[0]
        h, w = case["x"].shape[1:]
        stacked = MovementSimulator.stack_transformations(torch.from_numpy(case["mats"]), n // 2)
        assert_close(stacked.numpy(), case["stacked"], rtol=1e-6, atol=1e-6, what="stack_transformations")
        assert_exact(affines2theta(torch.from_numpy(case["stacked"]), h, w).numpy(), case["theta"], what="affine2theta")
        assert_exact(MovementSimulator.affine2theta(torch.from_numpy(case["stacked"][0]), h, w).numpy(), case["theta"][0], what="affine2theta (single)")


def test_random_affine_draw_order():
    """random_affine consumes numpy's global RNG like the reference: randint(size=2), uniform(size=2), uniform()."""
    from master_thesis_amd.utils.movement import MovementSimulator
    sim = MovementSimulator(10, 0.1, 0.2)
    np.random.seed(7)
    a = sim.random_affine().numpy()
    np.random.seed(7)
    tx, ty = np.random.randint(-10, 10, size=2)
    sx, sy = np.random.uniform(0.9, 1.1, size=2)
    rot = np.random.uniform(-0.2, 0.2)
    ref = np.array([[sx * np.cos(rot), -sy * np.sin(rot), tx], [sx * np.sin(rot), sy * np.cos(rot), ty], [0, 0, 1]]).astype(np.float32)
    assert_exact(a, ref)
    assert_exact(MovementSimulator.identity_affine().numpy(), np.eye(3, dtype=np.float32))


# ------------------------------------------------------------------------------------------------ GPU
class _AffineAligner:
    def align(self, x_target, m_target, x_refs, m_refs):
        from master_thesis_amd.models import cpn
        from master_thesis_amd import _ops
        b, c, r, h, w = x_refs.size()
        xa, va, _ = cpn.affine_align(x_refs, m_refs, None, standins.torch_theta(b * r, x_refs.device), list(range(r)))
        return xa, va, _ops.vmap(va, 1 - m_target)


def _aligner(name):
    return standins.torch_flow if name == "flow" else _AffineAligner()


def _run_product(alg, aligner, x, m, kw, batched=False):
    from master_thesis_amd import schedulers as S
    xt, mt = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
    suffix = "_batched" if batched else ""
    if alg == "cpn":
        y = getattr(S, "inpainting_algorithm_cpn" + suffix)(xt, mt, None, standins.torch_cpn_model)
    else:
        y = getattr(S, "inpainting_algorithm_%s%s" % (alg, suffix))(xt, mt, _aligner(aligner), standins.torch_inpaint, **kw)
    assert torch.equal(xt.cpu(), torch.from_numpy(x)) and torch.equal(mt.cpu(), torch.from_numpy(m)), "inputs were modified"
    return y.cpu().numpy()


@gpu
@pytest.mark.parametrize("case", load_golden("sequences"))
def test_schedulers_vs_reference_golden(case):
    f, h, w, hole, step = (int(v) for v in case["dims"])
    x, m = standins.moving_hole_sequence(int(case["seed"]), 3, f, h, w, hole, step)
    y = _run_product(str(case["alg"]), str(case["aligner"]), x, m, _kw(case))
    assert_close(y, case["y"], what="%s/%s" % (case["alg"], case["aligner"]))


@gpu
@pytest.mark.parametrize("alg,aligner,kw", [
    ("ff", "flow", dict(s=1, D=20, e=4)), ("ip", "flow", dict(s=1, D=3, e=4)), ("cp", "flow", dict(N=5, s=1, e=4)),
    ("cp", "affine", dict(N=4, s=2, e=2)), ("cpn", "none", dict())])
def test_batched_schedulers_vs_sequential_oracle(alg, aligner, kw):
    """Five sequences with different holes: they leave the active set at different slots."""
    f, h, w = 6, 16, 20
    seqs = [standins.moving_hole_sequence(100 + i, 3, f, h, w, hole, step) for i, (hole, step) in enumerate([(3, 1), (8, 1), (6, 3), (10, 2), (5, 0)])]
    seqs[0][1][:] = 0  # a sequence without any hole
    x, m = np.stack([s[0] for s in seqs]), np.stack([s[1] for s in seqs])
    y = _run_product(alg, aligner, x, m, kw, batched=True)
    for i in range(len(seqs)):
        assert_close(y[i], _oracle_sequence(alg, aligner, x[i], m[i], kw), what="%s sequence %d" % (alg, i))


@gpu
@pytest.mark.parametrize("case", load_golden("inpaint_step"))
def test_inpaint_runner_step_vs_reference_golden(case):
    from master_thesis_amd.runners import ThesisInpaintingRunner
    dev = lambda k: torch.from_numpy(case[k]).cuda()
    for fn in (ThesisInpaintingRunner.infer_step_propagate, ThesisInpaintingRunner.train_step_propagate):
        outs = fn(_aligner(str(case["aligner"])), standins.torch_inpaint, dev("x_t"), dev("m_t"), dev("x_r"), dev("m_r"))
        for got, key in zip(outs, ["x_ref_aligned", "v_ref_aligned", "v_map", "y_hat", "y_hat_comp"]):
            (assert_exact if key.startswith("v_") else assert_close)(got.cpu().numpy(), case[key], what="step " + key)


@gpu
def test_inpaint_compute_loss_matches_formula():
    from master_thesis_amd.runners import ThesisInpaintingRunner
    from master_thesis_amd.utils.losses import LossesUtils
    g = torch.Generator().manual_seed(3)
    b, f, h, w = 2, 1, 16, 16
    y_t, y_hat, y_comp = (torch.rand(*s, generator=g).cuda() for s in [(b, 3, h, w), (b, 3, f, h, w), (b, 3, f, h, w)])
    v_t = (torch.rand(b, 1, h, w, generator=g) < 0.6).float().cuda()
    v_map = ((torch.rand(b, 1, f, h, w, generator=g) < 0.5).float().cuda() * (1 - v_t).unsqueeze(2))
    vgg = lambda t: [t[:, :, ::2, ::2], t.mean(1, keepdim=True)]
    loss, items = ThesisInpaintingRunner.compute_loss(LossesUtils(vgg, "cuda"), y_t, v_t, y_hat, y_comp, v_map)
    tgt, nh = y_t.unsqueeze(2), v_t.unsqueeze(2)
    l1 = lambda a, mk, wgt: wgt * ((a - tgt) * mk).abs().sum() / (mk.sum() + 1e-9)
    ref = [l1(y_hat, nh.expand_as(v_map), 0.5), l1(y_hat, v_map, 2.0), l1(y_comp, (1 - nh) - v_map, 1.0)]
    for got, want in zip(items[:3], ref):
        assert abs(float(got) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    assert abs(float(loss) - float(sum(items))) < 1e-5


@gpu
@pytest.mark.parametrize("case", load_golden("movement"))
def test_movement_simulator_vs_reference_golden(case):
    from master_thesis_amd.utils.movement import MovementSimulator
    sim = MovementSimulator(10, 0.1, 0.2)
    x = torch.from_numpy(case["x"]).cuda()
    mats = [torch.from_numpy(a) for a in case["mats"]]
    y, flow_inv, back = sim.simulate_movement(x, len(mats), mats)
    assert back is mats
    assert_close(y.cpu().numpy(), case["y"], what="simulate_movement y")
    assert_close(flow_inv.cpu().numpy(), case["flow_inv"], rtol=1e-5, atol=2e-6, what="simulate_movement flow_inv")
    assert_close(MovementSimulator.transform_single(x, mats[0]).cpu().numpy(), case["single"], what="transform_single")


@gpu
def test_movement_simulator_batch_and_random():
    """B frames in one launch == B single calls; random matrices keep the middle frame untouched."""
    from master_thesis_amd.utils.movement import MovementSimulator
    sim = MovementSimulator(6, 0.1, 0.15)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(3, 3, 40, 56, generator=g).cuda()
    np.random.seed(5)
    y, flow_inv, mats = sim.simulate_movement_batch(x, 5)
    assert y.shape == (3, 3, 5, 40, 56) and flow_inv.shape == (3, 5, 40, 56, 2)
    assert_close(y[:, :, 2].cpu().numpy(), x.cpu().numpy(), atol=1e-5, what="identity middle frame")
    for b in range(3):
        y1, f1, _ = sim.simulate_movement(x[b], 5, mats[b])
        # batched vs single 3x3 LAPACK / BLAS calls on the host may differ in the last bit of theta
        assert_close(y1.cpu().numpy(), y[b].cpu().numpy(), atol=5e-5, what="single vs batch y")
        assert_close(f1.cpu().numpy(), flow_inv[b].cpu().numpy(), atol=5e-6, what="single vs batch flow_inv")
        yo, fo = SO.simulate_movement(x[b].cpu().numpy(), torch.stack(mats[b]).numpy())
        # the oracle inverts / chains the 3x3 matrices with numpy's LAPACK, the product with torch's (as the reference does):
        # last-bit differences in theta move sample positions by ~1e-5 px, hence the wider absolute floor here
        assert_close(y[b].cpu().numpy(), yo, atol=5e-5, what="batch y vs oracle")
        assert_close(flow_inv[b].cpu().numpy(), fo, atol=5e-6, what="batch flow_inv vs oracle")


@gpu
def test_affine_grid_matches_oracle_and_gradient():
    from master_thesis_amd import _ops
    from oracle import tcl_oracle as O
    g = torch.Generator().manual_seed(2)
    for n, h, w, ac in [(3, 17, 23, True), (2, 8, 8, False), (1, 1, 5, True), (4, 64, 48, False)]:
        theta = (torch.eye(2, 3).unsqueeze(0) + 0.2 * torch.randn(n, 2, 3, generator=g)).cuda().requires_grad_(True)
        grid = _ops.AffineGridFn.apply(theta, h, w, ac)
        assert_close(grid.detach().cpu().numpy(), O.affine_grid(theta.detach().cpu().numpy(), h, w, ac), rtol=1e-5, atol=2e-6, what="affine_grid")
        cot = torch.randn(n, h, w, 2, generator=g).cuda()
        (gt,) = torch.autograd.grad((grid * cot).sum(), theta)
        assert_close(gt.cpu().numpy(), O.affine_grid_bwd(cot.cpu().numpy(), h, w, ac), rtol=1e-4, atol=2e-5, what="affine_grid bwd")
