#!/usr/bin/env python3
"""Golden vectors for the rows SURVEY.md §8f marks "next": the sequence schedulers (N3), the inpainting runner's step
function (N2) and the movement simulator (N4), produced by running the REAL reference code.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden_sequences.py

The schedulers live in runner modules whose imports pull in packages this image lacks (skeltorch, progressbar,
torchvision, cv2, skimage, jpeg4py, tensorboardX).  None of them is touched by the functions captured here (pure torch
control flow + utils/flow.py), so they are satisfied by EMPTY auto-attribute modules; every reference module itself is
the real file, imported from /root/reference.  The networks the schedulers take as arguments are the weight-free
stand-ins of tests/standins.py (same functions on both sides of the comparison).  Fixtures are data only (.npz).
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import standins  # noqa: E402

REF = os.environ.get("TCL_REFERENCE", "/root/reference")
ABSENT = {"skeltorch", "progressbar", "torchvision", "cv2", "skimage", "jpeg4py", "tensorboardX", "lpips"}


class _Hollow(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


class _AbsentFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Hollow(spec.name)

    def exec_module(self, module):
        pass


def load_reference():
    sys.meta_path.insert(0, _AbsentFinder())
    for pkg in ["utils", "models", "thesis", "thesis_dfpn", "thesis_cpn", "thesis_inpainting"]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]  # bare package: __init__.py is NOT executed
        sys.modules[pkg] = m
    return (importlib.import_module("thesis_inpainting.runner"), importlib.import_module("thesis_cpn.runner"),
            importlib.import_module("utils.movement"), importlib.import_module("models.thesis_alignment"))


def save(name, cases):
    flat = {}
    for ci, case in enumerate(cases):
        for k, v in case.items():
            flat["c%d/%s" % (ci, k)] = np.asarray(v)
    flat["n_cases"] = np.asarray(len(cases))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print("%-28s %3d cases %8.1f KiB" % (name, len(cases), os.path.getsize(path) / 1024))


def make_schedules(inp, cpn):
    R, C = inp.ThesisInpaintingRunner, cpn.ThesisCPNRunner
    cases = []
    for max_t, s, D in [(1, 1, 20), (2, 1, 20), (7, 1, 20), (12, 2, 5), (30, 3, 7), (9, 1, 2)]:
        for t in range(max_t):
            cases.append(dict(kind="ff", args=[t, max_t, s, D], out=np.asarray(R.inpainting_algorithm_ff_indexes(t, max_t, s, D), dtype=np.int64)))
        t_list = sorted(list(range(max_t)), key=lambda xi: abs(xi - max_t // 2))
        for t in t_list:
            cases.append(dict(kind="ip", args=[t, max_t, s, D], out=np.asarray(R.inpainting_algorithm_ip_indexes(t, t_list, s, D), dtype=np.int64)))
    for n_frames, p, L in [(5, 2, 120), (50, 2, 120), (300, 2, 120), (300, 3, 40), (241, 2, 120), (10, 1, 3)]:
        for t in sorted({0, 1, n_frames // 3, n_frames // 2, n_frames - 2, n_frames - 1} & set(range(n_frames))):
            cases.append(dict(kind="cpn", args=[t, n_frames, p, L], out=np.asarray(C.get_reference_frame_indexes(t, n_frames, p, L), dtype=np.int64)))
    save("schedules", cases)


def make_sequences(inp, cpn, align_mod):
    R = inp.ThesisInpaintingRunner

    class FlowAligner(align_mod.ThesisAlignmentModel):  # passes the reference's isinstance dispatch; no weights
        def __init__(self):
            torch.nn.Module.__init__(self)

        def forward(self, x_target, m_target, x_refs, m_refs):
            return standins.torch_flow(x_target, m_target, x_refs, m_refs)

    class AffineAligner:  # the "else" branch: ThesisCPNRunner.infer_alignment_step_propagate -> model.align
        def align(self, x_target, m_target, x_refs, m_refs):
            import torch.nn.functional as F
            b, c, r, h, w = x_refs.size()
            grid = F.affine_grid(standins.torch_theta(b * r, x_refs.device), (b * r, c, h, w), align_corners=False)
            xa = F.grid_sample(x_refs.transpose(1, 2).reshape(-1, c, h, w), grid, align_corners=False).reshape(b, r, c, h, w).transpose(1, 2)
            va = (F.grid_sample(1 - m_refs.transpose(1, 2).reshape(-1, 1, h, w), grid, align_corners=False)
                  .reshape(b, r, 1, h, w).transpose(1, 2) > 0.5).float()
            return xa, va, (va - (1 - m_target.unsqueeze(2))).clamp(0, 1)

    cases = []
    specs = [
        # algorithm, aligner, seed, F, H, W, hole, step, kwargs
        ("ff", "flow", 0, 5, 16, 16, 6, 2, dict(s=1, D=20, e=1)),
        ("ff", "flow", 1, 6, 16, 20, 7, 1, dict(s=2, D=3, e=5)),
        ("ip", "flow", 2, 5, 16, 16, 6, 2, dict(s=1, D=20, e=1)),
        ("ip", "flow", 3, 7, 12, 16, 5, 1, dict(s=1, D=2, e=3)),
        ("cp", "flow", 4, 6, 16, 16, 6, 2, dict(N=4, s=1, e=1)),
        ("cp", "flow", 5, 8, 16, 16, 7, 1, dict(N=6, s=2, e=2)),
        ("cp", "affine", 6, 5, 16, 16, 6, 2, dict(N=4, s=1, e=1)),
        ("ff", "affine", 7, 4, 12, 12, 5, 2, dict(s=1, D=20, e=1)),
        ("cpn", "none", 8, 7, 10, 12, 4, 1, dict()),
    ]
    for alg, aligner, seed, f, h, w, hole, step, kw in specs:
        x, m = standins.moving_hole_sequence(seed, 3, f, h, w, hole, step)
        xt, mt = torch.from_numpy(x.copy()), torch.from_numpy(m.copy())  # IP / CP write into their arguments
        al = FlowAligner() if aligner == "flow" else AffineAligner()
        if alg == "cpn":
            y = cpn.ThesisCPNRunner.inpainting_algorithm(xt, mt, None, standins.torch_cpn_model)
        else:
            y = getattr(R, "inpainting_algorithm_" + alg)(xt, mt, al, standins.torch_inpaint, **kw)
        cases.append(dict(alg=alg, aligner=aligner, seed=seed, dims=[f, h, w, hole, step],
                          kw_keys=sorted(kw), kw_vals=[kw[k] for k in sorted(kw)], y=y.detach().numpy()))
    save("sequences", cases)

    # N2: one step of the inpainting runner (both aligner branches), all five outputs
    cases = []
    for aligner, seed in [("flow", 20), ("affine", 21)]:
        g = torch.Generator().manual_seed(seed)
        b, f, h, w = 2, 3, 16, 20
        x_t, x_r = torch.rand(b, 3, h, w, generator=g), torch.rand(b, 3, f, h, w, generator=g)
        m_t = (torch.rand(b, 1, h // 4, w // 4, generator=g) < 0.4).float().repeat_interleave(4, 2).repeat_interleave(4, 3)
        m_r = (torch.rand(b, 1, f, h // 4, w // 4, generator=g) < 0.3).float().repeat_interleave(4, 3).repeat_interleave(4, 4)
        al = FlowAligner() if aligner == "flow" else AffineAligner()
        outs = R.infer_step_propagate(al, standins.torch_inpaint, x_t, m_t, x_r, m_r)
        cases.append(dict(aligner=aligner, x_t=x_t.numpy(), m_t=m_t.numpy(), x_r=x_r.numpy(), m_r=m_r.numpy(),
                          x_ref_aligned=outs[0].numpy(), v_ref_aligned=outs[1].numpy(), v_map=outs[2].numpy(),
                          y_hat=outs[3].numpy(), y_hat_comp=outs[4].numpy()))
    save("inpaint_step", cases)


def affine_params(translation, scale, rotation):
    """What skimage.transform.AffineTransform(...).params holds (skimage is absent; published formula, shear = 0)."""
    (tx, ty), (sx, sy) = translation, scale
    return np.array([[sx * np.cos(rotation), -sy * np.sin(rotation), tx], [sx * np.sin(rotation), sy * np.cos(rotation), ty], [0, 0, 1]])


class _NumpyAsOfReference:
    """utils/movement.py:74 calls np.linalg.inv on a torch tensor and feeds the result to torch.from_numpy.  With the
    numpy / torch pair of the reference's time that returned an ndarray; numpy 2.x + torch 2.10 wrap it back into a Tensor
    and from_numpy raises.  This proxy (installed as the module's `np` for the generator run only) restores the ndarray
    return type; the arithmetic (LAPACK float32 inverse) is unchanged."""

    class linalg:
        @staticmethod
        def inv(a):
            return np.linalg.inv(np.asarray(a))

    def __getattr__(self, name):
        return getattr(np, name)


def make_movement(mov):
    mov.np = _NumpyAsOfReference()
    sim = mov.MovementSimulator(max_displacement=10, max_scaling=0.1, max_rotation=0.2)
    cases = []
    for seed, c, h, w, n in [(0, 3, 24, 32, 5), (1, 1, 16, 16, 4), (2, 3, 20, 12, 2), (3, 3, 17, 23, 7)]:
        rng = np.random.RandomState(40 + seed)
        x = torch.from_numpy(rng.rand(c, h, w).astype(np.float32))
        mats = []
        for i in range(n):
            if i == n // 2:
                mats.append(torch.eye(3))
            else:
                mats.append(torch.from_numpy(affine_params(rng.randint(-4, 4, size=2), rng.uniform(0.9, 1.1, size=2), rng.uniform(-0.2, 0.2))).float())
        y, flow_inv, _ = sim.simulate_movement(x, n, list(mats))
        single = mov.MovementSimulator.transform_single(x, mats[0])
        stacked = mov.MovementSimulator.stack_transformations(torch.stack(mats), n // 2)
        theta = torch.stack([mov.MovementSimulator.affine2theta(a, h, w) for a in stacked])
        cases.append(dict(x=x.numpy(), mats=torch.stack(mats).numpy(), y=y.numpy(), flow_inv=flow_inv.numpy(), single=single.numpy(),
                          stacked=stacked.numpy(), theta=theta.numpy()))
    save("movement", cases)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    inp, cpn, mov, align_mod = load_reference()
    make_schedules(inp, cpn)
    make_sequences(inp, cpn, align_mod)
    make_movement(mov)
    print("torch", torch.__version__)


if __name__ == "__main__":
    main()
