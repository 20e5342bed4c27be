"""reference src/inverter/utils_ae/util_pso_inverter.py:10-12: replace the assessor's
classifier head (2048 -> n_class, two classes by default: in / out of training)."""
import torch
from torch import nn


def change_classifier_n_class(model, n_class=2):
    device = model.fc.weight.device
    bound = 2048 ** -0.5  # nn.Linear's default initialisation range
    head = nn.Module()
    head.register_parameter("weight", nn.Parameter(torch.empty(n_class, 2048, device=device).uniform_(-bound, bound)))
    head.register_parameter("bias", nn.Parameter(torch.empty(n_class, device=device).uniform_(-bound, bound)))
    model._modules["fc"] = head
    model.n_class = n_class
    return model
