"""PSF surrogate network.  Only `MLP` is on the hot path (reference:
deeplens/psfnet_arch.py:24-47: 4 -> 64 -> 256 -> 8x256 -> ks^2, ReLU, Sigmoid, then
L1-normalise); the module/key layout `net.{0,2,...,20}.{weight,bias}` matches the
reference checkpoints so `load_state_dict` works on them unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as nnF


class MLP(nn.Module):
    def __init__(self, in_features, out_features, hidden_features=64, hidden_layers=3):
        super().__init__()
        layers = [nn.Linear(in_features, hidden_features // 4, bias=True), nn.ReLU(inplace=True),
                  nn.Linear(hidden_features // 4, hidden_features, bias=True), nn.ReLU(inplace=True)]
        for _ in range(hidden_layers):
            layers += [nn.Linear(hidden_features, hidden_features, bias=True), nn.ReLU(inplace=True)]
        layers += [nn.Linear(hidden_features, out_features, bias=True), nn.Sigmoid()]
        self.net = nn.Sequential(*layers)
        self.net.apply(initialize_weights)

    def forward(self, x):
        return nnF.normalize(self.net(x), p=1, dim=-1)


def initialize_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.kaiming_uniform_(m.weight.data)
        nn.init.constant_(m.bias.data, 0)
    elif isinstance(m, nn.Conv2d):
        nn.init.kaiming_uniform_(m.weight.data, nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0)
