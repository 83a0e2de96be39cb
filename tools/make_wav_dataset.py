"""Write a small synthetic WAV corpus + source CSVs in the reference's format (`filename,length.wav.samples`), to exercise the
real-data path of experiments/*.py without a downloaded dataset:  python tools/make_wav_dataset.py OUT_DIR [N] [MAX_SECONDS]"""
import math
import os
import sys
import wave

import numpy as np

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
max_s = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(0)
for split, count in (("train", n), ("test", max(n // 8, 2))):
    with open(os.path.join(out, f"{split}.csv"), "w") as f:
        f.write("filename,length.wav.samples\n")
        for i in range(count):
            T = int(16000 * max_s * rng.uniform(0.4, 1.0))
            t = np.arange(T) / 16000.0
            x = 0.4 * np.sin(2 * math.pi * rng.uniform(100, 800) * t) + 0.05 * rng.standard_normal(T)
            with wave.open(os.path.join(out, f"{split}_{i:04d}.wav"), "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(16000)
                w.writeframes((np.clip(x, -1, 1) * 32767).astype(np.int16).tobytes())
            f.write(f"{split}_{i:04d},{T}\n")
print(f"wrote {out}")
