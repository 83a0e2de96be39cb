"""BaseModel: init-kwarg capture and the three-file checkpoint format of the reference
(blvm/models/base_model.py:16-18 file names, :37-68 kwarg capture, :84-103 save/load)."""
import inspect
import logging
import os

import torch
import torch.nn as nn

LOGGER = logging.getLogger(name=__file__)

MODEL_CLASS_NAME_STR = "model_class_name.pt"
MODEL_INIT_KWRGS_STR = "model_kwargs.pt"
MODEL_STATE_DICT_STR = "model_state_dict.pt"


def load_model(path, model_class_name: str = None, device: torch.device = torch.device("cpu")):
    import blvm.models

    if not os.path.exists(path):
        raise RuntimeError(f"Tried to load model checkpoint but the path does not exist: {path}")
    if model_class_name is None:
        name_file = os.path.join(path, MODEL_CLASS_NAME_STR)
        if not os.path.exists(name_file):
            raise RuntimeError(f"Name of class of model to load was not given and not saved in checkpoint: {path}")
        model_class_name = torch.load(name_file, weights_only=True)
    return getattr(blvm.models, model_class_name).load(path, device=device)


class BaseModel(nn.Module):
    def __init__(self):
        super().__init__()
        self._init_arguments = None
        self._kwarg_names = [p for p in inspect.signature(self.__class__.__init__).parameters if p != "self"]

    def init_arguments(self):
        if self._init_arguments is None:
            missing = [n for n in self._kwarg_names if n not in vars(self)]
            if missing:
                LOGGER.warning(f"{self.__class__} does not keep these __init__ kwargs as attributes: {missing}")
            self._init_arguments = {a: getattr(self, a) for a in self._kwarg_names}
        return self._init_arguments

    @property
    def device(self):
        return next(self.parameters()).device

    def get_checkpoint(self):
        return dict(model_class_name=self.__class__.__name__, model_init_kwargs=self.init_arguments(),
                    model_state_dict=self.state_dict())  # fmt: skip

    def save(self, path):
        os.makedirs(path, exist_ok=True)
        torch.save(self.__class__.__name__, os.path.join(path, MODEL_CLASS_NAME_STR))
        torch.save(self.init_arguments(), os.path.join(path, MODEL_INIT_KWRGS_STR))
        torch.save(self.state_dict(), os.path.join(path, MODEL_STATE_DICT_STR))

    @classmethod
    def load(cls, path, device: str = "cpu"):
        kwargs = torch.load(os.path.join(path, MODEL_INIT_KWRGS_STR), weights_only=True)
        extra = kwargs.pop("kwargs", {})
        args = kwargs.pop("args", [])
        model = cls(*args, **extra, **kwargs)
        model.to(device)
        model.load_state_dict(torch.load(os.path.join(path, MODEL_STATE_DICT_STR), map_location=device, weights_only=True))
        return model

    def extra_repr(self):
        if not self.init_arguments():
            return ""
        s = ",\n  ".join(f"{k}={v}" for k, v in self.init_arguments().items() if not isinstance(v, nn.Module))
        return "kwargs={\n  " + s + "\n}"

    def summary(self, *args, **kwargs):
        n = sum(p.numel() for p in self.parameters())
        return f"{self.__class__.__name__}: {n:,} parameters"
