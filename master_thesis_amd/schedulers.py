"""Sequence schedulers of the reference, batched across sequences.

The reference inpaints ONE sequence at a time and calls the alignment + inpainting step once per (target frame,
reference frame) pair with batch size 1 (thesis_inpainting/runner.py:170-253, thesis_cpn/runner.py:121-171): thousands
of single-pair launches.  Which pair comes next depends only on the frame index and on how much of the target's hole is
still open, so S independent sequences of the same length walk the same schedule; they differ only in WHEN each one
stops.  The schedulers below therefore run S sequences in lock-step, one (S_active, ...) call per schedule slot, and
drop a sequence from the active set at the slot where the reference's own loop condition would have ended it.  Every
op inside a slot is per-sample, so each sequence's result is the one the reference's sequential loop gives.

x (S,C,F,H,W), m (S,1,F,H,W) -> y_inpainted (S,C,F,H,W).  The single-sequence functions keep the reference's
signature ((C,F,H,W) in, (C,F,H,W) out) and are the S=1 case.  Unlike the reference's IP / CP variants, the inputs are
not modified in place (the reference updates views of its arguments).

`step` is `runners.ThesisInpaintingRunner.infer_step_propagate`: the alignment network, `align_set` + `v_map` on the
HIP ops, then the inpainting network.
"""
import torch

from . import runners

FILL_COLOR = (0.485, 0.456, 0.406)


# ----------------------------------------------------------------------------------------------- schedules (pure host)
def inpainting_algorithm_ff_indexes(t, max_t, s, D):
    """Reference frames of target t, nearest first (ties: the earlier frame), within D frames, on the stride-s lattice.
    thesis_inpainting/runner.py:189-197."""
    others = [r for r in range(max_t) if r != t]
    others.sort(key=lambda r: (abs(r - t), r))
    return [r for r in others if abs(r - t) <= D and abs(r - t) % s == 0]


def inpainting_algorithm_ip_indexes(t, t_list, s, D):
    """Already-inpainted frames first (latest first), then the remaining FF candidates.  thesis_inpainting/runner.py:222-227."""
    done = t_list[:t_list.index(t)][::-1]
    return done + [r for r in inpainting_algorithm_ff_indexes(t, len(t_list), s, D) if r not in done]


def get_reference_frame_indexes(t, n_frames, p=2, r_list_max_length=120):
    """Every p-th frame of a window of +-r_list_max_length around t, shifted to stay inside the sequence.
    thesis_cpn/runner.py:154-171."""
    start, end = t - r_list_max_length, t + r_list_max_length
    if start < 0:
        end = end - start
        end = n_frames - 1 if end > n_frames else end
        start = 0
    elif end > n_frames:
        start = start - (end - n_frames)
        start = 0 if start < 0 else start
        end = n_frames - 1
    return [i for i in range(start, end, p) if i != t]


# ----------------------------------------------------------------------------------------------- helpers
def _fill(x):
    return torch.as_tensor(FILL_COLOR, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)


def _coverage(m_t):
    """Percentage of hole pixels per sequence, as python floats (one host sync per schedule slot, as in the reference)."""
    per = m_t[0].numel()
    return (m_t.reshape(m_t.size(0), -1).sum(1) * 100 / per).tolist()


def _step(model_alignment, model, idx, x_t, m_t, x_r, m_r):
    """One schedule slot on the sequences `idx` (list of ints).  Returns (v_map[:, :, 0], y_hat_comp[:, :, 0])."""
    if len(idx) != x_t.size(0):
        sel = torch.as_tensor(idx, device=x_t.device)
        x_t, m_t, x_r, m_r = x_t.index_select(0, sel), m_t.index_select(0, sel), x_r.index_select(0, sel), m_r.index_select(0, sel)
    _, _, v_map, _, y_hat_comp = runners.ThesisInpaintingRunner.infer_step_propagate(model_alignment, model, x_t, m_t, x_r, m_r)
    return v_map[:, :, 0], y_hat_comp[:, :, 0]


def _put(dst, idx, val):
    if len(idx) == dst.size(0):
        dst.copy_(val)
    else:
        dst.index_copy_(0, torch.as_tensor(idx, device=dst.device), val)


def _take(src, idx):
    return src if len(idx) == src.size(0) else src.index_select(0, torch.as_tensor(idx, device=src.device))


# ----------------------------------------------------------------------------------------------- FF
def inpainting_algorithm_ff_batched(x, m, model_alignment, model, s=1, D=20, e=1):
    """Frame-by-frame: every target starts from the ORIGINAL frames and pulls from its references, nearest first, until
    less than e % of it is hole.  thesis_inpainting/runner.py:170-187."""
    S, c, f, h, w = x.size()
    fill = _fill(x)
    y_inpainted = torch.zeros_like(x)
    for t in range(f):
        x_target, m_target = x[:, :, t].clone(), m[:, :, t].clone()
        last = torch.zeros_like(x_target)
        cands = inpainting_algorithm_ff_indexes(t, f, s=s, D=D)
        active, first = list(range(S)), True
        while active and (first or cands):
            r = [cands.pop(0)]
            v_map, comp = _step(model_alignment, model, active, x_target, m_target, x[:, :, r], m[:, :, r])
            m_new = _take(m_target, active) - v_map
            _put(m_target, active, m_new)
            _put(x_target, active, (1 - m_new) * comp + m_new.repeat(1, 3, 1, 1) * fill)
            _put(last, active, comp)
            cov = _coverage(m_new)
            active, first = [i for i, cv in zip(active, cov) if cv > e], False
        y_inpainted[:, :, t] = last
    return y_inpainted


def inpainting_algorithm_ff(x, m, model_alignment, model, s=1, D=20, e=1):
    return inpainting_algorithm_ff_batched(x.unsqueeze(0), m.unsqueeze(0), model_alignment, model, s=s, D=D, e=e)[0]


# ----------------------------------------------------------------------------------------------- IP
def inpainting_algorithm_ip_batched(x, m, model_alignment, model, s=1, D=20, e=1):
    """Inpaint-and-propagate: targets from the middle of the sequence outwards; finished frames become references.
    thesis_inpainting/runner.py:199-220."""
    S, c, f, h, w = x.size()
    fill = _fill(x)
    y_inp, m_inp = x.clone(), m.clone()
    t_list = sorted(range(f), key=lambda ti: abs(ti - f // 2))
    for t in t_list:
        cands = inpainting_algorithm_ip_indexes(t, t_list, s, D)
        last = torch.zeros_like(y_inp[:, :, t])
        active, first = list(range(S)), True
        while active and (first or cands):
            r = [cands.pop(0)]
            v_map, comp = _step(model_alignment, model, active, y_inp[:, :, t], m_inp[:, :, t], y_inp[:, :, r], m_inp[:, :, r])
            m_new = _take(m_inp[:, :, t], active) - v_map
            _put(m_inp[:, :, t], active, m_new)
            _put(y_inp[:, :, t], active, (1 - m_new) * comp + m_new.repeat(1, 3, 1, 1) * fill)
            _put(last, active, comp)
            cov = _coverage(m_new)
            active, first = [i for i, cv in zip(active, cov) if cv > e], False
        m_inp[:, :, t] = 0
        y_inp[:, :, t] = last
    return y_inp


def inpainting_algorithm_ip(x, m, model_alignment, model, s=1, D=20, e=1):
    return inpainting_algorithm_ip_batched(x.unsqueeze(0), m.unsqueeze(0), model_alignment, model, s=s, D=D, e=e)[0]


# ----------------------------------------------------------------------------------------------- CP
def inpainting_algorithm_cp_batched(x, m, model_alignment, model, N=20, s=1, e=1):
    """Copy-and-propagate: N sweeps; on alternating sweeps the even / odd frames pull from their two neighbours at
    distance s; a frame is closed when less than e % of it is hole, or in the last two sweeps.
    thesis_inpainting/runner.py:229-253."""
    S, c, f, h, w = x.size()
    fill = _fill(x)
    y_inp, m_inp = x.clone(), m.clone()
    for i in range(N):
        t_list = [t for t in range(f) if (t // s) % (s if s > 1 else 2) == i % 2]
        for t in t_list:
            open_px = m_inp[:, :, t].reshape(S, -1).sum(1).tolist()
            active = [k for k in range(S) if open_px[k] != 0]
            if not active:
                continue
            for delta_t in (-s, s):
                if not 0 <= t + delta_t < f:
                    continue
                r = [t + delta_t]
                v_map, comp = _step(model_alignment, model, active, y_inp[:, :, t], m_inp[:, :, t], y_inp[:, :, r], m_inp[:, :, r])
                m_new = _take(m_inp[:, :, t], active) - v_map
                y_new = (1 - m_new) * comp + m_new.repeat(1, 3, 1, 1) * fill
                close = torch.as_tensor([cv < e or i >= N - 2 for cv in _coverage(m_new)], device=x.device).view(-1, 1, 1, 1)
                m_new = torch.where(close, torch.zeros_like(m_new), m_new)
                y_new = torch.where(close, comp, y_new)
                _put(m_inp[:, :, t], active, m_new)
                _put(y_inp[:, :, t], active, y_new)
    return y_inp


def inpainting_algorithm_cp(x, m, model_alignment, model, N=20, s=1, e=1):
    return inpainting_algorithm_cp_batched(x.unsqueeze(0), m.unsqueeze(0), model_alignment, model, N=N, s=s, e=e)[0]


# ----------------------------------------------------------------------------------------------- CPN bidirectional
def inpainting_algorithm_cpn_batched(x, m, model_alignment, model):
    """The original CPN schedule: one pass forwards and one backwards over the frames, each frame inpainted from every
    second frame of the (partly inpainted) sequence and written back; the two passes are blended linearly in time.
    thesis_cpn/runner.py:121-152.  `model(x_t, m_t, x_refs, m_refs) -> (y_hat, y_hat_comp)`; model_alignment is unused."""
    S, c, f, h, w = x.size()
    passes = []
    for d in range(2):
        x_copy, m_copy = x.clone(), m.clone()
        y_d = torch.zeros_like(x)
        for t in (range(f) if d == 0 else range(f - 1, -1, -1)):
            r_list = get_reference_frame_indexes(t, f)
            with torch.no_grad():
                _, comp = model(x_copy[:, :, t], m_copy[:, :, t], x_copy[:, :, r_list], m_copy[:, :, r_list])
            x_copy[:, :, t] = comp
            m_copy[:, :, t] = 0
            y_d[:, :, t] = comp
        passes.append(y_d)
    k = torch.arange(f, dtype=torch.float64, device=x.device).view(1, 1, f, 1, 1)
    return (passes[0].double() * (k / f) + passes[1].double() * ((f - k) / f)).float()


def inpainting_algorithm_cpn(x, m, model_alignment, model):
    return inpainting_algorithm_cpn_batched(x.unsqueeze(0), m.unsqueeze(0), model_alignment, model)[0]
