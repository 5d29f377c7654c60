"""Base class of AppZoo applications -- mirror of the reference contract
(easynlp/appzoo/application.py:26-38): an ``nn.Module`` with ``forward(inputs)``
and ``compute_loss(forward_outputs, label_ids, **kwargs) -> {'loss': Tensor}``."""
import torch.nn as nn


class Application(nn.Module):

    def __init__(self):
        super().__init__()

    def init_weights(self):
        raise NotImplementedError

    def forward(self, inputs):
        raise NotImplementedError

    def compute_loss(self, forward_outputs, label_ids, **kwargs):
        raise NotImplementedError

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        return cls(pretrained_model_name_or_path, **kwargs)
