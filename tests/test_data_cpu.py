"""CPU: the host-side data front-end against fixtures produced by the imported reference
(tests/golden/data.npz, oracle/gen_golden.py::gen_data): length-bucketed samplers under a seeded `random`, padded
collation, the source-CSV dataset on WAV files, rank sharding of batches."""
import os
import random
import wave

import numpy as np
import torch

from blvm.data.base_dataset import BaseDataset
from blvm.data.batchers import DynamicTensorBatcher
from blvm.data.loaders import AudioLoader
from blvm.data.samplers import LengthEvalSampler, LengthTrainSampler
from blvm.data.transforms import Compose, Denormalize, MuLawDecode, MuLawEncode, Normalize, RandomSegment

from conftest import GOLDEN


def _split(flat, sizes):
    out, i = [], 0
    for n in sizes:
        out.append([int(v) for v in flat[i : i + n]])
        i += n
    return out


def test_length_train_sampler_matches_reference():
    g = np.load(os.path.join(GOLDEN, "data.npz"))
    lengths = g["lengths"].tolist()
    random.seed(7)
    s = LengthTrainSampler(lengths, batch_len=16000 * 20, min_pool_size=64, max_pool_difference=4000.0)
    assert [list(map(int, p)) for p in s.pools] == _split(g["train_pools_flat"], g["train_pool_sizes"])
    for ep in range(2):
        got = [list(map(int, b)) for b in iter(s)]
        assert got == _split(g[f"train_ep{ep}_flat"], g[f"train_ep{ep}_sizes"])
        assert all(sum(lengths[i] for i in b) <= 16000 * 20 for b in got)
    random.seed(9)
    s = LengthTrainSampler(lengths, batch_len="2max", min_pool_size=128, num_batches=11, drop_last=False, longest_first=False)
    # `num_batches`: the reference draws the same epoch but then overwrites the selection with the whole epoch
    # (`self.batches = batches` after the recursive call, length_samplers.py:185-189); here exactly num_batches are served —
    # the first num_batches of the same shuffled epoch
    got = [list(map(int, b)) for b in iter(s)]
    assert len(got) == 11 and got == _split(g["train_nb_flat"], g["train_nb_sizes"])[:11]


def test_length_eval_sampler_matches_reference_and_shards():
    g = np.load(os.path.join(GOLDEN, "data.npz"))
    lengths = g["lengths"].tolist()
    for tag, kw in (("len", dict(batch_len=16000 * 30)), ("size", dict(batch_size=32))):
        got = [list(map(int, b)) for b in iter(LengthEvalSampler(lengths, **kw))]
        assert got == _split(g[f"eval_{tag}_flat"], g[f"eval_{tag}_sizes"])
    # rank sharding: the union over ranks is the global batch, no example twice
    full = [list(map(int, b)) for b in iter(LengthEvalSampler(lengths, batch_size=32))]
    parts = [[list(map(int, b)) for b in iter(LengthEvalSampler(lengths, batch_size=32, rank=r, world_size=4))] for r in range(4)]
    for i, b in enumerate(full):
        assert sorted(sum((parts[r][i] for r in range(4)), [])) == sorted(b)


def test_dynamic_tensor_batcher_matches_reference():
    g = np.load(os.path.join(GOLDEN, "data.npz"))
    flat, xs = torch.from_numpy(g["collate_in"]), []
    for n in g["collate_lens"]:
        xs.append(flat[: int(n)].clone())
        flat = flat[int(n) :]
    out, sl = DynamicTensorBatcher().collate(xs)
    assert torch.equal(out, torch.from_numpy(g["collate_out"])) and torch.equal(sl, torch.from_numpy(g["collate_sl"]))
    flat, xs2 = torch.from_numpy(g["collate2_in"]), []
    for n in (7, 4, 9):
        xs2.append(flat[: 3 * n].view(3, n).clone())
        flat = flat[3 * n :]
    out2, sl2 = DynamicTensorBatcher(dim=-1, pad_value=-1.0).collate(xs2)
    assert torch.equal(out2, torch.from_numpy(g["collate2_out"])) and torch.equal(sl2, torch.from_numpy(g["collate2_sl"]))


def test_source_csv_dataset_on_wav_files(tmp_path):
    rate, lens = 16000, [1600, 900, 2400]
    gen = torch.Generator().manual_seed(0)
    with open(tmp_path / "train.csv", "w") as f:
        f.write("filename,length.wav.samples\n")
        for i, n in enumerate(lens):
            pcm = (torch.rand(n, generator=gen) * 2 - 1).mul(20000).to(torch.int16).numpy()
            with wave.open(str(tmp_path / f"utt{i}.wav"), "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate)  # noqa: E702
                w.writeframes(pcm.tobytes())
            f.write(f"utt{i},{n}\n")
    enc = Compose(MuLawEncode(bits=16))
    ds = BaseDataset(str(tmp_path / "train.csv"), modalities=[(AudioLoader("wav"), enc, DynamicTensorBatcher())])
    sampler = LengthEvalSampler(str(tmp_path / "train.csv"), field="length", batch_size=3)
    loader = torch.utils.data.DataLoader(ds, batch_sampler=sampler, collate_fn=ds.collate)
    (x, x_sl), meta = next(iter(loader))
    assert x.shape == (3, 2400) and x_sl.tolist() == [2400, 1600, 900]  # longest first, right zero padding
    assert float(x.abs().max()) <= 1.0 and float(x[2, 900:].abs().max()) == 0.0
    assert [m["sample_rate"] for m in meta] == [rate] * 3
    torch.testing.assert_close(MuLawDecode(16)(MuLawEncode(16)(x)), x, rtol=1e-4, atol=1e-6)
    torch.manual_seed(1)
    seg = RandomSegment(500)(x[0])
    assert seg.shape == (500,)
    n = Normalize(mean=0.1, std=2.0)
    torch.testing.assert_close(Denormalize(mean=0.1, std=2.0)(n(x)), x)
