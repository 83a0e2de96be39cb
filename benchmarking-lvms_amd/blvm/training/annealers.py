"""Host-side schedules for beta / free nats (reference: blvm/training/annealers.py:21-72)."""
import math


class Annealer:
    value = None

    def step(self) -> float:
        raise NotImplementedError()


class CosineAnnealer(Annealer):
    """`value` is None until the first `step()`; constant for `constant_steps`, then half a cosine period from
    `start_value` to `end_value` over `anneal_steps` calls."""

    def __init__(self, anneal_steps: int, constant_steps: int = 0, start_value: float = 0, end_value: float = 1):
        if anneal_steps < 0 or constant_steps < 0:
            raise ValueError(f"steps must be positive but got {anneal_steps=}, {constant_steps=}")
        if not math.isfinite(start_value) or not math.isfinite(end_value):
            raise ValueError(f"start_value and end_value must be finite but got {start_value=}, {end_value=}")
        self.anneal_steps, self.constant_steps = anneal_steps, constant_steps
        self.start_value, self.end_value = start_value, end_value
        self.steps = 0
        self.value = None

    def step(self):
        self.steps += 1
        k = self.steps
        if k >= self.anneal_steps + self.constant_steps:
            self.value = self.end_value
        elif k <= self.constant_steps:
            self.value = self.start_value
        else:
            phase = (k - self.constant_steps - 1) / self.anneal_steps * math.pi
            self.value = self.end_value + 0.5 * (self.start_value - self.end_value) * (1 + math.cos(phase))
        return self.value

    def __repr__(self):
        return (f"CosineAnnealer(anneal_steps={self.anneal_steps}, constant_steps={self.constant_steps} "
                f"start_value={self.start_value}, end_value={self.end_value})")
