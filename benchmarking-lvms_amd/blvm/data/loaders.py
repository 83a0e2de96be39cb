"""File loaders with the reference's names (blvm/data/loaders.py:145-162 AudioLoader).  WAV files are decoded with the
standard library; other containers (FLAC) need torchaudio, which is used when importable and reported clearly when not."""
import wave

import numpy as np
import torch


class Loader:
    def __init__(self, extension=None, cache: bool = False):
        self.extension = extension.strip(".") if extension else None
        self.cache = cache
        self._cache = {}

    def __call__(self, example_id: str):
        path = example_id if self.extension is None or example_id.endswith("." + self.extension) else f"{example_id}.{self.extension}"
        if self.cache and path in self._cache:
            return self._cache[path]
        out = self.load(path)
        if self.cache:
            self._cache[path] = out
        return out

    def load(self, path):
        raise NotImplementedError()


class AudioLoader(Loader):
    """Waveform [T] float32 in [-1, 1] (channels summed, loaders.py:48-53) and its metadata."""

    def __init__(self, extension: str = "wav", cache: bool = False, sum_channels: bool = True):
        super().__init__(extension, cache)
        self.sum_channels = sum_channels

    def load(self, path: str):
        if path.lower().endswith(".wav"):
            with wave.open(path, "rb") as f:
                n_ch, width, rate, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
                raw = f.readframes(n)
            if width not in (1, 2, 4):
                raise ValueError(f"{path}: unsupported sample width {width}")
            dtype, scale = {1: (np.uint8, 128.0), 2: (np.int16, 32768.0), 4: (np.int32, 2147483648.0)}[width]
            x = np.frombuffer(raw, dtype=dtype).astype(np.float32)
            x = (x - 128.0) / scale if width == 1 else x / scale
            x = torch.from_numpy(x.reshape(-1, n_ch).T.copy())  # [channels, T]
        else:
            try:
                import torchaudio
            except ImportError as e:
                raise ImportError(f"{path}: decoding this container needs torchaudio, which is not installed; convert to WAV") from e
            x, rate = torchaudio.load(path)
        x = x.sum(0) if self.sum_channels else x
        return x, dict(sample_rate=rate, length=x.shape[-1], file=path)
