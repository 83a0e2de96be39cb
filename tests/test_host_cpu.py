"""CPU: host-side logic of the product (no GPU compute): library exports, construction/initialisation parity with
the reference, state_dict layout, shape helpers, metrics, annealer, and the loud failure on CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import blvm_oracle as O
from blvm import _hip
from blvm.evaluation import BitsPerDimMetric, DeferredScalars, KLMetric, LLMetric, LossMetric, Tracker
from blvm.models import VRNNAudio
from blvm.training.annealers import CosineAnnealer
from blvm.utils import operations as OP

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "blvm_hip.h")).read()
    declared = set(re.findall(r"\b(blvm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_hip.EXPORTS)
    lib = ctypes.CDLL(_hip.lib_path())  # loads without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert _hip.load().blvm_version() >= 100


def test_cpu_tensor_is_refused_loudly():
    m = VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16)
    x, x_sl = O.synth_batch(2, 32, seed=1)
    with pytest.raises(_hip.BlvmHipError, match="no CPU fallback"):
        m(x, x_sl)


def test_grouped_weight_gradient_refuses_cpu_and_misshapen_operands():
    """`ops.wgrad_group` (one launch for the weight gradients of a chain) checks its operands on the host before any pointer reaches
    the library: CPU tensors, non-unit column strides and row counts that differ from `rows` raise; an empty job list is a no-op."""
    from blvm import ops

    ops.wgrad_group([], 16)
    ops.wgrad_group([(torch.zeros(16, 4), torch.zeros(16, 4), None, None)], 16)  # nothing wanted: nothing launched
    D, X, dW = torch.zeros(16, 4), torch.zeros(16, 8), torch.zeros(4, 8)
    with pytest.raises(_hip.BlvmHipError, match="HIP tensors"):
        ops.wgrad_group([(D, X, dW, None)], 16)
    with pytest.raises(_hip.BlvmHipError, match=r"rows=32"):
        ops.wgrad_group([(D, X, dW, None)], 32)


def test_vrnn_audio_init_and_state_dict_match_reference():
    """Same seed + same construction order => identical parameters as the reference (pinned by checksums)."""
    g = np.load(os.path.join(GOLDEN, "vrnn_full.npz"))
    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, num_mix=10, num_bins=2**16)
    sd = m.state_dict()
    assert list(sd.keys()) == g["param_names"].tolist()
    for k, v in sd.items():
        cks = g[f"cks.{k}"]
        assert list(v.shape) == [int(s) for s in cks[2:]], k
        assert v.double().sum().item() == pytest.approx(cks[0], rel=1e-12, abs=1e-9), k
        assert v.double().abs().sum().item() == pytest.approx(cks[1], rel=1e-12), k


def test_oracle_full_size_against_reference_outputs():
    """Full C2 dimensions: oracle (fed with the product's seeded parameters) reproduces the reference's outputs."""
    g = np.load(os.path.join(GOLDEN, "vrnn_full.npz"))
    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True)
    sd = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, x_sl = O.synth_batch(4, 1280, seed=0, ragged=True)
    assert x_sl.tolist() == g["x_sl"].tolist()
    assert x.double().sum().item() == pytest.approx(g["x_cks"][0], rel=1e-12)
    torch.manual_seed(123)
    eps = torch.stack([torch.randn(4, 256) for _ in range(20)], 0)
    assert eps.double().abs().sum().item() == pytest.approx(g["eps_cks"][1], rel=1e-12)
    out = O.vrnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=2.0, stack=64)
    np.testing.assert_allclose(out["loss"].item(), g["loss"], rtol=1e-7)
    np.testing.assert_allclose(out["elbo"].detach().numpy(), g["elbo"], rtol=1e-7)
    np.testing.assert_allclose(out["kl"].detach().numpy(), g["kl"], rtol=1e-6)
    out["loss"].backward()
    for name, ref in zip(g["grad_names"].tolist(), g["grad_norms"].tolist()):
        assert sd[name].grad.double().norm().item() == pytest.approx(ref, rel=2e-4), name


def test_operations_match_oracle_and_reference_vectors():
    fn = np.load(os.path.join(GOLDEN, "functions.npz"))
    st, pad = OP.stack_tensor(torch.from_numpy(fn["stack_x"]), 8, dim=1)
    assert pad == int(fn["stack_pad"]) and torch.equal(st, torch.from_numpy(fn["stack_out"]))
    # like the reference, unstacking [B,T',S] keeps a trailing singleton dim: [B,T,1] (operations.py:44-47)
    assert torch.equal(OP.unstack_tensor(st, 8, pad, dim=-1).squeeze(-1), torch.from_numpy(fn["stack_x"]))
    sl = torch.from_numpy(fn["mask_sl"])
    assert torch.equal(OP.sequence_mask(sl), torch.from_numpy(fn["mask_bool"]))
    assert torch.equal(OP.reverse_sequences(torch.from_numpy(fn["rev_x"]), sl), torch.from_numpy(fn["rev_out"]))
    # split_sequence: consume / extend bookkeeping
    x = torch.arange(2 * 10).view(2, 10).float()
    xs, sls = OP.split_sequence(x, torch.tensor([10, 6]), length=4, overlap=0)
    assert [tuple(t.shape) for t in xs] == [(2, 4), (2, 4), (1, 2)] and [s.tolist() for s in sls] == [[4, 4], [4, 2], [2]]
    with pytest.raises(ValueError):
        OP.split_sequence(x, torch.tensor([10, 6]), length=4, overlap=4)


def test_annealer_matches_reference_trace_and_contract():
    fn = np.load(os.path.join(GOLDEN, "functions.npz"))
    a = CosineAnnealer(anneal_steps=10, constant_steps=0, start_value=0, end_value=1)
    assert a.value is None  # None before the first step (reference test_annealers.py)
    np.testing.assert_allclose([a.step() for _ in range(15)], fn["anneal_beta"], atol=1e-15)
    b = CosineAnnealer(anneal_steps=7, constant_steps=5, start_value=2.0, end_value=0.0)
    tr = [b.step() for _ in range(15)]
    np.testing.assert_allclose(tr, fn["anneal_fn"], atol=1e-15)
    assert all(x > y for x, y in zip(tr[5:11], tr[6:12]))  # strictly monotone while annealing
    assert tr[-1] == 0.0
    for bad in [dict(anneal_steps=-1), dict(anneal_steps=1, constant_steps=-2), dict(anneal_steps=1, start_value=float("inf"))]:
        with pytest.raises(ValueError):
            CosineAnnealer(**bad)


def test_metrics_merge_like_reference():
    elbo1, sl1 = torch.tensor([-100.0, -50.0]), torch.tensor([10, 5])
    elbo2, sl2 = torch.tensor([-30.0]), torch.tensor([3])
    m1, m2 = BitsPerDimMetric(elbo1, reduce_by=sl1), BitsPerDimMetric(elbo2, reduce_by=sl2)
    assert m1.value == pytest.approx(150 / np.log(2) / 15)
    m1.update(m2)
    assert m1.value == pytest.approx(180 / np.log(2) / 18) and m1.weight_by == 18
    # deferred device scalars behave like the tensors they summarise
    d = DeferredScalars(torch.tensor([1.5, -150.0]))
    assert LossMetric(d[0], weight_by=2).value == 1.5
    assert LLMetric(d[1], name="elbo", reduce_by=2).value == -75.0
    assert KLMetric(d[1] / np.log(2), name="kl (bpt)", reduce_by=15).value == pytest.approx(-150 / np.log(2) / 15)
    t = Tracker()
    t.update([LLMetric(elbo1, name="elbo")], source="train")
    t.update([LLMetric(elbo2, name="elbo")], source="train")
    assert t.values("train")["elbo"] == pytest.approx(-60.0)


def test_lstm_audio_init_matches_reference_and_oracle_c1():
    """BASELINE config C1 (`experiment_lstm_audio.py`, [8,4000], CPU reference): seeded init identical to the
    reference, and the oracle reproduces the reference's loss / ll / gradient norms on full and ragged batches."""
    from blvm.models import LSTMAudio

    g = np.load(os.path.join(GOLDEN, "lstm.npz"))
    torch.manual_seed(0)
    m = LSTMAudio(stack_size=64, hidden_size=256, num_layers=1, num_mix=10, num_bins=2**16)
    sd0 = m.state_dict()
    assert list(sd0.keys()) == g["param_names"].tolist()
    for k, v in sd0.items():
        cks = g[f"cks.{k}"]
        assert list(v.shape) == [int(s) for s in cks[2:]], k
        assert v.double().abs().sum().item() == pytest.approx(cks[1], rel=1e-12), k
    for tag, ragged in (("full", False), ("ragged", True)):
        sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        x, x_sl = O.synth_batch(8, 4000, seed=0, ragged=ragged)
        assert x_sl.tolist() == g[f"{tag}_x_sl"].tolist()
        out = O.lstm_audio_forward(sd, x, x_sl, stack=64, num_bins=2**16)
        np.testing.assert_allclose(out["loss"].item(), g[f"{tag}_loss"], rtol=1e-6)
        np.testing.assert_allclose(out["ll"].detach().numpy(), g[f"{tag}_ll"], rtol=1e-6)
        out["loss"].backward()
        for name, ref in zip(g["grad_names"].tolist(), g[f"{tag}_grad_norms"].tolist()):
            assert sd[name].grad.double().norm().item() == pytest.approx(ref, rel=2e-4), name


def test_srnn_audio_init_matches_reference_and_oracle_full_dims():
    from blvm.models import SRNNAudio

    g = np.load(os.path.join(GOLDEN, "srnn.npz"))
    torch.manual_seed(0)
    m = SRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, smoothing=True)
    sd0 = m.state_dict()
    assert list(sd0.keys()) == g["param_names"].tolist()
    for k, v in sd0.items():
        cks = g[f"cks.{k}"]
        assert list(v.shape) == [int(s) for s in cks[2:]], k
        assert v.double().abs().sum().item() == pytest.approx(cks[1], rel=1e-12), k
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    x, x_sl = O.synth_batch(4, 1280, seed=0, ragged=True)
    torch.manual_seed(123)
    eps = torch.stack([torch.randn(4, 256) for _ in range(20)], 0)
    out = O.srnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=2.0, stack=64)
    np.testing.assert_allclose(out["loss"].item(), g["f_loss"], rtol=1e-7)
    np.testing.assert_allclose(out["elbo"].detach().numpy(), g["f_elbo"], rtol=1e-7)
    out["loss"].backward()
    for name, ref in zip(g["f_grad_names"].tolist(), g["f_grad_norms"].tolist()):
        assert sd[name].grad.double().norm().item() == pytest.approx(ref, rel=2e-4), name


def test_padding_helpers():
    from blvm.utils.padding import get_modulo_length, get_modulo_padding, get_same_padding

    assert get_modulo_length(1000, 64, kernel_size=64) == 1024 and get_modulo_length(1024, 64, 64) == 1024
    assert get_modulo_padding(10, 4, 2) == 0 and get_modulo_padding(11, 4, 2) == 3
    assert get_same_padding(16000, 64, 64) == 0 and get_same_padding(16001, 64, 64) == 63
    with pytest.raises(ValueError):
        get_modulo_padding(3, 4, 8)


def test_wavenet_init_matches_reference_and_oracle_c5_dims():
    from blvm.models import WaveNet
    from blvm.modules.distributions import DiscretizedLogisticMixtureDense

    g = np.load(os.path.join(GOLDEN, "wavenet.npz"))
    torch.manual_seed(0)
    lik = DiscretizedLogisticMixtureDense(96, 1, num_mix=10, num_bins=2**16)
    m = WaveNet(likelihood=lik, n_layers=10, n_stacks=5, res_channels=96, kernel_size=2, base_dilation=2, n_stack_frames=1)
    assert m.receptive_field == int(g["f_rf"]) == 5117
    sd0 = m.state_dict()
    assert list(sd0.keys()) == g["param_names"].tolist()
    for k, v in sd0.items():
        cks = g[f"cks.{k}"]
        assert list(v.shape) == [int(s) for s in cks[2:]], k
        assert v.double().abs().sum().item() == pytest.approx(cks[1], rel=1e-12), k
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    x, x_sl = O.synth_batch(2, 1500, seed=0, ragged=True)
    out = O.wavenet_forward(sd, x, x_sl, n_layers=10, n_stacks=5)
    np.testing.assert_allclose(out["loss"].item(), g["f_loss"], rtol=2e-6)
    np.testing.assert_allclose(out["log_prob"].detach().numpy(), g["f_log_prob"], rtol=2e-6)
    out["loss"].backward()
    for name, ref in zip(g["f_grad_names"].tolist(), g["f_grad_norms"].tolist()):
        assert sd[name].grad.double().norm().item() == pytest.approx(ref, rel=5e-4), name


def test_host_helper_modules_the_entry_points_import(tmp_path):
    """The small host-side modules the reference's experiment scripts import resolve here too (argparsing, rand, device,
    optimization, restore, data registry), and a run checkpoint round-trips."""
    import torch

    from blvm.data import BaseDataset  # noqa: F401
    from blvm.data.datasets import DATASETS
    from blvm.models import LSTMAudio
    from blvm.training.restore import load_run, save_run
    from blvm.utils.argparsing import str2bool
    from blvm.utils.device import get_device  # noqa: F401
    from blvm.utils.optimization import get_learning_rates_dict
    from blvm.utils.rand import get_random_seed, set_seed

    assert str2bool("true") is True and "timit" in DATASETS and DATASETS["timit"].audio_ext == "flac"
    set_seed(5)
    a = torch.rand(3)
    set_seed(5)
    assert torch.equal(a, torch.rand(3)) and 0 <= get_random_seed() < 2**32
    m = LSTMAudio(stack_size=8, hidden_size=16, num_layers=1, num_mix=10, num_bins=256)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    assert get_learning_rates_dict(opt) == {"lr": 1e-3}
    sched.step()
    save_run(str(tmp_path), model=m, optimizer=opt, lr_scheduler=sched)
    m2 = LSTMAudio(stack_size=8, hidden_size=16, num_layers=1, num_mix=10, num_bins=256)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, milestones=[1], gamma=0.1)
    _, ck = load_run(str(tmp_path), model=m2, optimizer=opt2, lr_scheduler=sched2)
    assert all(torch.equal(p, q) for p, q in zip(m.state_dict().values(), m2.state_dict().values()))
    assert get_learning_rates_dict(opt2)["lr"] == get_learning_rates_dict(opt)["lr"] and "optimizer_state_dict" in ck


def test_persistent_chain_tile_iterator_covers_every_tile_once(tmp_path):
    """Host-side check of the placement logic of the persistent kernels (csrc/pchain.h TileIter): for plain and XCD-aware
    placement, every (row tile, column tile) of a link is visited exactly once and only by workgroups of the link's range — a
    workgroup outside the range must see no tile (an out-of-range tile index is an out-of-bounds access on the GPU)."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "tile_iter_test"
    src = os.path.join(ROOT, "tests", "host", "tile_iter_test.hip")
    inc = [f"-I{os.path.join(ROOT, 'include')}", f"-I{os.path.join(ROOT, 'benchmarking-lvms_amd', 'csrc')}"]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-w", *inc, src, "-o", str(exe)], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "0 errors" in out.stdout, out.stdout + out.stderr


def test_wavenet_stack_buffer_layout_is_host_arithmetic():
    """`blvm_wavenet_stack_floats` (the sizes of the two buffers `blvm_wavenet_stack_fwd / _bwd` slice block outputs and reserves
    from) against the per-block sizes it is defined by; a dilation that leaves no output frames is refused.  No GPU call."""
    lib = _hip.load()
    L, B, C = 900, 3, 32
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    arr = (ctypes.c_int * len(dil))(*dil)
    na, nr = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.blvm_wavenet_stack_floats(L, B, C, arr, len(dil), ctypes.byref(na), ctypes.byref(nr)) == 0
    Li, acts, res = L, 0, 0
    for i, d in enumerate(dil):
        res += lib.blvm_wavenet_block_reserve_floats(Li, B, C, d)
        if i + 1 < len(dil):
            acts += (Li - d) * B * C
        Li -= d
    assert (na.value, nr.value) == (acts, res)
    assert na.value % 4 == 0 and nr.value % 4 == 0  # every slice starts 16-byte aligned
    bad = (ctypes.c_int * 2)(5, 900)
    assert lib.blvm_wavenet_stack_floats(L, B, C, bad, 2, ctypes.byref(na), ctypes.byref(nr)) != 0


def test_save_run_load_run_round_trip_with_the_weights_only_loader(tmp_path):
    """checkpoints are read back with loaders that execute nothing from the file."""
    from blvm.training.restore import load_run, save_run

    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.5)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    sched.step()
    from types import SimpleNamespace

    save_run(str(tmp_path), m, opt, sched, tracker=SimpleNamespace(epoch=3))
    torch.manual_seed(1)
    m2 = VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, milestones=[2], gamma=0.5)
    m2, ck = load_run(str(tmp_path), m2, opt2, sched2)
    assert ck["epoch"] == 3
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]
    m3 = type(m).load(str(tmp_path))  # BaseModel.load: class name + kwargs + state dict, all through weights_only loads
    assert torch.equal(m3.state_dict()["vrnn.encoder.2.weight"], m.state_dict()["vrnn.encoder.2.weight"])
