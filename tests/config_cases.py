"""Seeded inputs at the BASELINE.json config shapes + the checksum format of tests/golden/configs.npz.

Shared by tests/golden/make_golden_configs.py (which runs the REAL reference on these inputs in the build container and
stores checksums of its outputs) and by the tests (which regenerate the same inputs from the seed, run the oracle / the HIP
path and compare against the stored checksums).  Inputs come from numpy's PCG64 (`default_rng`), which is bit-reproducible
across machines, so only outputs have to be stored, and only as checksums:

    float tensors: sum, abs-sum, 32 contiguous-chunk sums (+ their abs-sums, the error scale) and 64 sampled values
    masks        : exact popcount, a 64-bit BLAKE2b digest of the packed bits, and the packed bits themselves
                   (they compress to a few KiB: block masks) so that a mismatch can be located

SURVEY.md §8c: "one medium case per config shape stored as checksums".
"""
import hashlib

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------------------------------------ checksums
def float_checksum(a, n_samples=64, n_chunks=32):
    a = np.ascontiguousarray(a)
    flat = a.reshape(-1).astype(np.float64)
    idx = np.random.default_rng(a.size).integers(0, a.size, n_samples)
    parts = np.array_split(flat, n_chunks)
    return {
        "shape": np.asarray(a.shape, dtype=np.int64),
        "sum": np.float64(flat.sum()),
        "abs_sum": np.float64(np.abs(flat).sum()),
        "chunk_sums": np.array([p.sum() for p in parts]),
        "chunk_abs": np.array([np.abs(p).sum() for p in parts]),
        "idx": idx.astype(np.int64),
        "samples": a.reshape(-1)[idx].astype(np.float32),
    }


def mask_checksum(a):
    bits = np.packbits(np.ascontiguousarray(a).reshape(-1) != 0)
    return {
        "shape": np.asarray(a.shape, dtype=np.int64),
        "popcount": np.int64(np.count_nonzero(a)),
        "digest": np.frombuffer(hashlib.blake2b(bits.tobytes(), digest_size=8).digest(), dtype=np.uint8).copy(),
        "bits": bits,
    }


def check_float(got, ref, rtol=1e-4, atol=1e-5, what=""):
    """`got` (array) against a stored float_checksum `ref` (dict of arrays).  Samples: the fp32 bar element-wise.  Chunk sums:
    |difference| <= rtol * (chunk's abs-sum) + atol * elements: what element-wise errors of that size can add up to."""
    got = np.ascontiguousarray(got)
    assert tuple(got.shape) == tuple(int(s) for s in ref["shape"]), "%s: shape %s vs %s" % (what, got.shape, ref["shape"])
    flat = got.reshape(-1)
    samples = flat[ref["idx"]].astype(np.float64)
    want = ref["samples"].astype(np.float64)
    scale = max(1.0, float(np.max(np.abs(want))))
    bad = np.abs(samples - want) > atol * scale + rtol * np.abs(want)
    assert not bad.any(), "%s: %d/%d sampled values off, e.g. got %r want %r" % (what, bad.sum(), bad.size, samples[bad][0], want[bad][0])
    parts = np.array_split(flat.astype(np.float64), len(ref["chunk_sums"]))
    sums = np.array([p.sum() for p in parts])
    sizes = np.array([p.size for p in parts])
    tol = rtol * ref["chunk_abs"] + atol * sizes * 1e-2
    bad = np.abs(sums - ref["chunk_sums"]) > tol
    assert not bad.any(), "%s: chunk sums off in chunks %s (got %r want %r)" % (what, np.nonzero(bad)[0][:4], sums[bad][:2], ref["chunk_sums"][bad][:2])
    assert abs(float(flat.astype(np.float64).sum()) - float(ref["sum"])) <= rtol * float(ref["abs_sum"]) + atol * flat.size * 1e-2, "%s: sum" % what


def check_mask(got, ref, what=""):
    """Bit-exact: popcount, digest, and (to locate a mismatch) the bits."""
    got = np.ascontiguousarray(got)
    assert tuple(got.shape) == tuple(int(s) for s in ref["shape"]), "%s: shape" % what
    assert set(np.unique(got)) <= {0.0, 1.0}, "%s: not a {0,1} mask" % what
    bits = np.packbits(got.reshape(-1) != 0)
    if not np.array_equal(bits, ref["bits"]):
        diff = np.unpackbits(bits ^ ref["bits"])[:got.size].reshape(got.shape)
        where = np.argwhere(diff)
        raise AssertionError("%s: %d mask elements differ from the reference's, first at %s" % (what, len(where), where[:5].tolist()))
    assert int(np.count_nonzero(got)) == int(ref["popcount"])
    digest = np.frombuffer(hashlib.blake2b(bits.tobytes(), digest_size=8).digest(), dtype=np.uint8)
    assert np.array_equal(digest, ref["digest"]), "%s: digest" % what


# ------------------------------------------------------------------------------------------------ inputs
def _blocky(rng, shape_coarse, cell, p):
    """{0,1} float map made of cell x cell blocks over the last two axes; P(1) = p."""
    m = (rng.random(shape_coarse) < p).astype(F32)
    return np.repeat(np.repeat(m, cell, axis=-2), cell, axis=-1)


def corr_c2():
    """One pair of C2: 64 x 64 maps, C = 256, masks with 10 % zeros, random cotangent."""
    rng = np.random.default_rng(2002)
    B, C, F_, S = 1, 256, 1, 64
    return dict(x_t=rng.standard_normal((B, C, S, S)).astype(F32), x_r=rng.standard_normal((B, C, F_, S, S)).astype(F32),
                v_t=(rng.random((B, 1, S, S)) >= 0.1).astype(F32), v_r=(rng.random((B, 1, F_, S, S)) >= 0.1).astype(F32),
                cot=rng.standard_normal((B, F_, S, S, S, S)).astype(F32))


def corr_c2_batch():
    """configs[1] at its full batch: 16 pairs of 64 x 64 maps, C = 256 (a 1 GiB volume and a 1 GiB cotangent)."""
    rng = np.random.default_rng(2016)
    B, C, F_, S = 16, 256, 1, 64
    return dict(x_t=rng.standard_normal((B, C, S, S), dtype=F32), x_r=rng.standard_normal((B, C, F_, S, S), dtype=F32),
                v_t=(rng.random((B, 1, S, S)) >= 0.1).astype(F32), v_r=(rng.random((B, 1, F_, S, S)) >= 0.1).astype(F32),
                cot=rng.standard_normal((B, F_, S, S, S, S), dtype=F32))


def corr_c3():
    """The alignment model's shape (C1 / C3): 16 x 16 x 512, 4 references, post-ReLU-like features incl. all-zero pixels."""
    rng = np.random.default_rng(2003)
    B, C, F_, S = 1, 512, 4, 16
    x_t = np.maximum(rng.standard_normal((B, C, S, S)), 0).astype(F32)
    x_r = np.maximum(rng.standard_normal((B, C, F_, S, S)), 0).astype(F32)
    x_t[:, :, 3, 5] = 0
    x_r[:, :, 1, 7, 2] = 0
    return dict(x_t=x_t, x_r=x_r, v_t=_blocky(rng, (B, 1, S // 2, S // 2), 2, 0.85), v_r=_blocky(rng, (B, 1, F_, S // 2, S // 2), 2, 0.85),
                cot=rng.standard_normal((B, F_, S, S, S, S)).astype(F32))


def _affine_flow(rng, n, H, W, max_shift_px=50.0, max_scale=0.1, max_rot=0.2, jitter=0.002):
    """identity grid under a random similarity (translation <= 50 px, scale +-10 %, rotation +-0.2 rad; cf. gts_movement_params,
    config.alignment.json:21) plus a small per-pixel jitter: ~5-10 % of the samples fall outside the frame."""
    xs, ys = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    out = np.empty((n, H, W, 2), F32)
    for i in range(n):
        s = 1 + rng.uniform(-max_scale, max_scale)
        a = rng.uniform(-max_rot, max_rot)
        tx, ty = rng.uniform(-1, 1, 2) * max_shift_px * 2 / np.array([W, H])
        gx = s * (np.cos(a) * xs - np.sin(a) * ys) + tx
        gy = s * (np.sin(a) * xs + np.cos(a) * ys) + ty
        out[i] = np.stack([gx, gy], -1) + jitter * rng.standard_normal((H, W, 2))
    return out


def align_c3():
    """One sequence of C3: 4 references at 256 x 256, `align_set` with a random-similarity flow."""
    rng = np.random.default_rng(3003)
    B, F_, S = 1, 4, 256
    return dict(x=rng.random((B, 3, F_, S, S)).astype(F32), v=_blocky(rng, (B, 1, F_, S // 8, S // 8), 8, 0.8),
                flow=_affine_flow(rng, B * F_, S, S).reshape(B, F_, S, S, 2), cot=rng.standard_normal((B, 3, F_, S, S)).astype(F32))


def cpn_align_c4():
    """One target of C4: 240 x 424 frames, 10 references, theta = identity + N(0, 0.05)."""
    rng = np.random.default_rng(4004)
    B, R, H, W = 1, 10, 240, 424
    F_ = R + 1
    t = F_ // 2
    r_list = [i for i in range(F_) if i != t]
    theta = (np.array([[1, 0, 0], [0, 1, 0]], F32)[None] + 0.05 * rng.standard_normal((B * R, 2, 3))).astype(F32)
    return dict(x=rng.random((B, 3, F_, H, W)).astype(F32), y=rng.random((B, 3, F_, H, W)).astype(F32),
                m=1 - _blocky(rng, (B, 1, F_, H // 8, W // 8), 8, 0.7), theta=theta, t=t, r_list=r_list,
                cot=rng.standard_normal((B, 3, R, H, W)).astype(F32))


CTX_VARIANTS = [("default", False), ("default", True), ("complete", True), ("cm_module", True)]


def ctx_c4(normalize):
    """One target of C4: c_feats (1,128,11,30,53), masks at 240 x 424.  Un-normalised similarities (how CPNet calls it,
    models/cpn.py:319) get small features so that exp() stays in range."""
    rng = np.random.default_rng(4040)
    B, C, R, h, w, H, W = 1, 128, 10, 30, 53, 240, 424
    scale = 1.0 if normalize else 0.05
    cf = (scale * rng.standard_normal((B, C, R + 1, h, w))).astype(F32)
    v_t = _blocky(rng, (B, 1, H // 4, W // 4), 4, 0.6)
    v_al = _blocky(rng, (B, 1, R, H // 4, W // 4), 4, 0.7)
    v_al[0, :, 3] = 0                      # a reference with zero overlap
    v_al[:, :, :, :16, :24] = 0            # pixels visible in no reference
    return dict(c_feats=cf, v_t=v_t, v_aligned=v_al)


def ctx_cots(p_in_shape, c_mask_shape):
    rng = np.random.default_rng(4041)
    return rng.standard_normal(p_in_shape).astype(F32), rng.standard_normal(c_mask_shape).astype(F32)


def loss_c3():
    """Arguments of ThesisAlignmentRunner.compute_loss (thesis_dfpn/runner.py:162-206) at the training shapes: one sequence of
    5 frames at 256 x 256, t = 2."""
    rng = np.random.default_rng(5005)
    B, F_, t = 2, 5, 2
    r_list = [0, 1, 3, 4]
    R = len(r_list)
    d = dict(t=t, r_list=r_list, flows_use=np.array([True, False]))
    d["corr"] = rng.uniform(0, 1, (B, R, 16, 16, 16, 16)).astype(F32)
    for k, s in enumerate((16, 64, 256)):
        d["x%d" % k] = rng.random((B, 3, F_, s, s)).astype(F32)
        d["v%d" % k] = _blocky(rng, (B, 1, F_, s // 4, s // 4), 4, 0.8)
        d["y%d" % k] = rng.random((B, 3, F_, s, s)).astype(F32)
        d["xa%d" % k] = rng.random((B, 3, R, s, s)).astype(F32)
        d["flow%d" % k] = _affine_flow(rng, B * R, s, s, max_shift_px=0.2 * s).reshape(B, R, s, s, 2)
        d["flow_gt%d" % k] = _affine_flow(rng, B * R, s, s, max_shift_px=0.2 * s).reshape(B, R, s, s, 2)
    return d
