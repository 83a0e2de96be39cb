from .length_samplers import LengthEvalSampler, LengthTrainSampler  # noqa: F401
