"""GPU (through the C ABI): split evaluation against the reference's own split-evaluation loops
(tests/golden/split_eval.npz, oracle/gen_golden.py::gen_split_eval), and WaveNet at BASELINE configs[4]'s own shape [4,1,16000].

WaveNet: `split_sequence` + `forward_split` as driven by the entry point's loop (experiments/_common.wavenet_split_eval =
experiment_wavenet_audio.py:224-231), both split modes.  STCN: `forward_split` for the first and a later split.  CW-VAE: the one case
the reference's loop completes (an utterance batch that fits one split) and the IndexError it raises on every split that is not the
last (`pad_same=False`; 98 of 98 probed shapes in the fixture)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

import blvm_oracle as O
from blvm import _hip
from blvm.evaluation import Tracker

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "experiments"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_hip():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert _hip.load().blvm_device_ok() == 1, "libblvm_hip: no gfx950 device visible"


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "split_eval.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol, atol=0.0):
    torch.testing.assert_close(a.detach().double().cpu(), (b if isinstance(b, torch.Tensor) else T(b)).double(), rtol=rtol, atol=atol)


def _wavenet(g):
    from blvm.models import WaveNet
    from blvm.modules.distributions import DiscretizedLogisticMixtureDense

    m = WaveNet(likelihood=DiscretizedLogisticMixtureDense(16, 1, num_mix=10, num_bins=2**16), n_layers=3, n_stacks=2, res_channels=16,
                kernel_size=2, base_dilation=2, n_stack_frames=1)  # fmt: skip
    m.load_state_dict({k[6:]: T(g[k]) for k in g.files if k.startswith("wn_sd.")})
    return m.to(DEV).eval()


@pytest.mark.parametrize("tag", ["consume", "extend"])
def test_wavenet_split_evaluation_matches_the_reference_loop(g, tag):
    """Per split: the product's splits are the reference's, loss / per-utterance log-prob / per-frame log-prob / the three metrics are
    the reference's ("extend": splits whose loss is 0, negative, or 83 nats per frame because the length the loss is normalised by is
    reduced by the receptive field — kept).  Then the entry point's loop on the whole batch: merged tracker values."""
    import _common as C

    m = _wavenet(g)
    x, x_sl, length = T(g["wn_x"]).to(DEV), T(g["wn_x_sl"]), int(g[f"wn_{tag}_length"])
    sd64 = {k[6:]: T(g[k]).double() for k in g.files if k.startswith("wn_sd.")}

    def near(a, ref, truth, rtol, atol):
        """|a - ref| within rtol / atol, or a as close to the float64 evaluation as the reference's own fp32 result is (x2): the
        reference evaluates the DMoL bin mass as a difference of two fp32 sigmoids 2^-16 apart, the kernel cancellation-free — on
        the "extend" splits every scored frame is the same zero-padded frame, so that rounding does not average out."""
        a, ref, truth = a.detach().double().cpu(), T(ref).double(), truth.double()
        tol = torch.maximum(atol + rtol * ref.abs(), 2 * (ref - truth).abs() + atol)
        assert ((a - ref).abs() <= tol).all() or ((a - truth).abs() <= tol).all(), (a, ref, truth)

    with torch.no_grad():
        xs, sls = m.split_sequence(x, x_sl, length=length)
        assert len(xs) == int(g[f"wn_{tag}_n"])
        for i, (x_i, sl_i) in enumerate(zip(xs, sls)):
            assert torch.equal(x_i.cpu(), T(g[f"wn_{tag}_x{i}"])) and torch.equal(sl_i, T(g[f"wn_{tag}_x_sl{i}"])), i
            loss, metrics, out = m.forward_split(x_i.contiguous(), sl_i, i_split=i)
            t64 = O.wavenet_forward(sd64, x_i.double().cpu(), sl_i, n_layers=3, n_stacks=2, pad_causal=True, pad_receptive_field=(i == 0))
            near(loss, g[f"wn_{tag}_loss{i}"], t64["loss"], 1e-5, 1e-5)
            near(out.log_prob, g[f"wn_{tag}_log_prob{i}"], t64["log_prob"], 1e-5, 1e-3)
            near(out.log_prob_twise, g[f"wn_{tag}_ll_twise{i}"], t64["log_prob_twise"], 1e-4, 1e-4)
            assert [mm.name for mm in metrics] == list(g[f"wn_{tag}_metric_names"])
            np.testing.assert_allclose([mm.value for mm in metrics], g[f"wn_{tag}_metric_values{i}"], rtol=5e-5, atol=1e-5)
        tracker = Tracker()
        tracker.source = "test"
        C.wavenet_split_eval(m, x, x_sl, tracker, length)
    torch.cuda.synchronize()
    merged = tracker.values("test")
    assert list(merged) == list(g[f"wn_{tag}_merged_names"])
    np.testing.assert_allclose(list(merged.values()), g[f"wn_{tag}_merged_values"], rtol=5e-5, atol=1e-5)


@pytest.mark.parametrize("i_split", [0, 1])
def test_stcn_forward_split_matches_reference(g, i_split):
    from blvm.models import STCN

    m = STCN(likelihood="DMoL", n_layers=3, latent_size=[16, 16, 32], res_channels=16, n_stack_frames=8)
    sd = {k[6:]: T(g[k]) for k in g.files if k.startswith("st_sd.")}
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with pytest.raises(NotImplementedError):  # stcn.py:328-330
        m.split_sequence(T(g["st_x"]).to(DEV), T(g["st_x_sl"]), length=100)
    eps = [T(g[f"st_eps{i_split}_{l}"]).transpose(0, 1).contiguous().to(DEV) for l in range(3)]
    with torch.no_grad():
        loss, metrics, o = m.forward_split(T(g["st_x"]).to(DEV), T(g["st_x_sl"]), i_split=i_split, eps=eps)
    close(loss, g[f"st_loss{i_split}"], 1e-4)
    close(o.elbo, g[f"st_elbo{i_split}"], 1e-4, 1e-3)
    close(o.log_prob, g[f"st_log_prob{i_split}"], 1e-4, 1e-3)
    for l in range(3):
        close(o.z[l], g[f"st_z{i_split}_{l}"], 1e-4, 1e-4)
        close(o.klds[l], g[f"st_kld{i_split}_{l}"], 1e-4, 1e-4)
    assert [mm.name for mm in metrics] == list(g[f"st_metric_names{i_split}"])
    np.testing.assert_allclose([mm.value for mm in metrics], g[f"st_metric_values{i_split}"], rtol=1e-4, atol=1e-5)


def _cwvae():
    from blvm.models import CWVAEAudio

    c = np.load(os.path.join(GOLDEN, "cwvae.npz"))  # the reduced model of gen_cwvae: gen_split_eval builds it from the same seed
    m = CWVAEAudio(z_size=[32, 16, 16], h_size=16, strides=[4, 2, 2], num_level_layers=2, stride_per_layer=2, likelihood="DMoL",
                   num_mix=10, num_bins=2**16, precision_posterior=True)  # fmt: skip
    m.load_state_dict({k[6:]: T(c[k]) for k in c.files if k.startswith("pw_sd.")})
    return m.to(DEV).eval()


def test_cwvae_split_evaluation_single_split_matches_reference(g):
    """`split_sequence` at a length the whole batch fits -> one split -> `forward_split(is_last_split=True)`: same padding, the
    reference's loss / ELBO / KL / latents / carried states; then the entry point's loop (device noise): it completes and its
    merged "rec" metrics — which do not depend on the noise — are the reference's."""
    import _common as C

    m = _cwvae()
    x, x_sl = T(g["cw_x"]).to(DEV), T(g["cw_x_sl"])
    xs, sls = m.split_sequence(x, x_sl, length=1024)
    assert len(xs) == 1 and torch.equal(sls[0], T(g["cw_one_x_sl"]))
    eps = [T(g[f"cw_one_eps{l}"]).to(DEV) for l in range(3)]
    with torch.no_grad():
        loss, metrics, o = m.forward_split(xs[0], sls[0], is_last_split=True, eps=eps)
        close(loss, g["cw_one_loss"], 1e-4)
        close(o.elbo, g["cw_one_elbo"], 1e-4)
        close(o.log_prob, g["cw_one_log_prob"], 1e-4)
        close(o.kld, g["cw_one_kld"], 1e-4, 1e-4)
        for l in range(3):
            close(o.z[l], g[f"cw_one_z{l}"], 1e-4, 1e-4)
            close(o.state_n[l][0], g[f"cw_one_state_z{l}"], 1e-4, 1e-4)
            close(o.state_n[l][1], g[f"cw_one_state_h{l}"], 1e-4, 1e-4)
        ref = dict(zip(g["cw_one_merged_names"].tolist(), g["cw_one_merged_values"].tolist()))
        vals = {mm.name: mm.value for mm in metrics}
        assert list(vals) == list(ref)
        for k, v in ref.items():
            assert vals[k] == pytest.approx(v, rel=1e-4, abs=1e-6), k
        tracker = Tracker()
        tracker.source = "test"
        C.cwvae_split_eval(m, x, x_sl, tracker, 1024)
    torch.cuda.synchronize()
    _hip.check_async()
    merged = tracker.values("test")
    assert list(merged) == list(ref) and all(math.isfinite(v) for v in merged.values())


@pytest.mark.parametrize("length", [256, 300])
def test_cwvae_split_that_is_not_the_last_raises_like_the_reference(g, length):
    """`forward(pad_same=False)` raises IndexError in the reference for every shape (fixture: 98 of 98) because the lengths are not
    reduced by what the un-padded convolutions consume (quirk 8) — the product raises the same error before launching anything, and
    the entry point's loop therefore stops at the first of several splits, as the reference's does."""
    import _common as C

    assert list(g["cw_not_last_raises"]) == ["IndexError"]
    m = _cwvae()
    x, x_sl = T(g["cw_x"]).to(DEV), T(g["cw_x_sl"])
    xs, sls = m.split_sequence(x, x_sl, length=length)
    assert [list(t.shape) for t in xs] == g[f"cw_split{length}_shapes"].tolist()
    assert torch.equal(torch.stack(sls), T(g[f"cw_split{length}_x_sl"]))
    with torch.no_grad():
        with pytest.raises(IndexError):
            m.forward_split(xs[0].contiguous(), sls[0], is_last_split=False)
        with pytest.raises(IndexError):
            C.cwvae_split_eval(m, x, x_sl, Tracker(), length)
        for L in (157, 205, 269, 301, 477):  # the fixture's sweep, sampled
            with pytest.raises(IndexError):
                m.forward_split(x[:, :L].contiguous(), torch.tensor([L, L, L]), is_last_split=False)
        loss, _, _ = m.forward_split(xs[-1].contiguous(), sls[-1], is_last_split=True)  # the last split alone is an ordinary forward
    assert math.isfinite(float(loss))


# ---- BASELINE configs[4] at its own shape: WaveNet 5 x 10, C = 96, [4,1,16000] ------------------------------------------------------
def _c5():
    from blvm.models import WaveNet
    from blvm.modules.distributions import DiscretizedLogisticMixtureDense

    torch.manual_seed(0)
    lik = DiscretizedLogisticMixtureDense(96, 1, num_mix=10, num_bins=2**16)
    return WaveNet(likelihood=lik, n_layers=10, n_stacks=5, res_channels=96, kernel_size=2, base_dilation=2, n_stack_frames=1).to(DEV)


def _grads(m, x, x_sl):
    m.zero_grad(set_to_none=True)
    loss, metrics, out = m(x, x_sl)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), out.log_prob.detach().cpu(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, metrics


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_wavenet_c5_at_its_own_shape(monkeypatch):
    """[4,1,16000] (experiments/benchmarks.txt:7, BASELINE configs[4]): 84 468 rows per block — the production row count of the
    column-split weight-gradient kernel.  (i) bits/dim at random init in the window every model starts in; (ii) rows
    are independent: utterance b of the batch = the same utterance alone; (iii) samples beyond x_sl are inert; (iv) the
    column-split weight-gradient form against the whole-output form on the same inputs (same sums in another order); (v) the
    bf16-operand mode (the reference runs this config under autocast) within the A.4 budget of the fp32 step."""
    m = _c5()
    B, T_ = 4, 16000
    x, _ = O.synth_batch(B, T_, seed=0)
    x = x.to(DEV)
    full = torch.full((B,), T_, dtype=torch.int64)
    loss, lp, grads, metrics = _grads(m, x, full)
    bpd = {mm.name: mm.value for mm in metrics}["bpd"]
    assert 16.5 < bpd < 18.0, bpd  # log2(65536) + ~1 at random init
    assert bpd == pytest.approx(loss / math.log(2), rel=1e-5)
    # (ii) row independence, forward: one utterance alone
    with torch.no_grad():
        _, _, o1 = m(x[2:3].contiguous(), full[2:3])
    assert float(o1.log_prob[0]) == pytest.approx(float(lp[2]), rel=2e-6)
    # (iii) ragged lengths: garbage beyond x_sl changes nothing, neither in the loss nor in any gradient
    x_sl = torch.tensor([16000, 12345, 9000, 5117])
    la, lpa, ga, _ = _grads(m, x, x_sl)
    xg = x.clone()
    for b, n in enumerate(x_sl.tolist()):
        xg[b, n:] = 0.77
    lb, lpb, gb, _ = _grads(m, xg, x_sl)
    assert la == pytest.approx(lb, rel=1e-6)  # (sums by atomics: the order, not the terms, may differ between two runs)
    torch.testing.assert_close(lpa, lpb, rtol=1e-6, atol=0)
    for k in ga:
        assert rel_l2(ga[k], gb[k]) < 1e-5, k
    # the first utterance is full length in both runs: its log-prob does not move with the others' lengths
    assert float(lpa[0]) == pytest.approx(float(lp[0]), rel=2e-6)
    # (iv) weight-gradient kernel: column-split form (what this row count selects) vs the whole-output form
    monkeypatch.setenv("BLVM_WN_WGRAD_NPW", "6")  # NPW = C / 16: every workgroup owns the whole output
    _, _, g_whole, _ = _grads(m, x, x_sl)
    monkeypatch.setenv("BLVM_WN_WGRAD_NPW", "1")  # one column tile per workgroup
    _, _, g_one, _ = _grads(m, x, x_sl)
    monkeypatch.delenv("BLVM_WN_WGRAD_NPW")
    worst = 0.0
    for k in ga:
        if "res_blocks" in k:  # fp32 atomics in a different order: round-off of sums over 6e4 rows
            worst = max(worst, rel_l2(ga[k], g_whole[k]), rel_l2(g_one[k], g_whole[k]))
    assert worst < 2e-5, worst
    # (v) bf16 operands / fp32 accumulation at this shape
    try:
        _hip.set_operand_dtype("bf16")
        l16, lp16, g16, _ = _grads(m, x, x_sl)
    finally:
        _hip.set_operand_dtype("f32")
    assert abs(l16 - la) / abs(la) < 1e-4  # 1e-4 relative on the per-frame log-likelihood
    torch.testing.assert_close(lp16, lpa, rtol=1e-4, atol=0)
    num = sum(float((g16[k].double() * ga[k].double()).sum()) for k in ga)
    den = math.sqrt(sum(float(g16[k].double().pow(2).sum()) for k in ga) * sum(float(ga[k].double().pow(2).sum()) for k in ga))
    assert num / den > 0.995, num / den
    _hip.check_async()
