"""MI355X-native twins of the reference's CNN networks: ``QNetwork`` -- the critic the
shipped configuration trains with (``critic_type: "CNN"``, config.yaml:61; got_sac_network.py:125-170) -- and the
CNN actor ``GaussianPolicy`` (:258-327).  Same constructor signatures, attribute names and ``state_dict`` keys
(``conv1..3`` keep the reference's (cout, cin, 5, 5) weight layout); the conv stack is one fused HIP node
(functional.cnn_features), every Linear runs on the MFMA GEMM.
"""
import torch
from torch import nn
from torch.distributions import Normal

from . import functional as F_
from . import sac_networks as _S
from .sac_networks import LOG_SIG_MAX, LOG_SIG_MIN, epsilon, weights_init_, _lin, _head, _action_affine


class _ConvStack(nn.Module):
    def _make_convs(self):
        self.conv1 = nn.Conv2d(1, 16, 5, stride=2)
        self.conv2 = nn.Conv2d(16, 64, 5, stride=2)
        self.conv3 = nn.Conv2d(64, 256, 5, stride=2)
        self.avg = nn.AdaptiveAvgPool2d(output_size=(1, 1))

    def _features(self, istate):
        return F_.cnn_features(istate, [self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                        self.conv3.weight, self.conv3.bias])


class QNetwork(_ConvStack):
    """Twin-Q CNN critic: forward([istate (B,H,W), pstate (B,2), a (B,2)]) -> (q1, q2)."""

    def __init__(self, nb_actions, nb_pstate):
        super().__init__()
        self._make_convs()
        self.fc1 = nn.Linear(256 + 32 + nb_actions, 128)
        self.fc2 = nn.Linear(128, 32)
        self.fc3 = nn.Linear(32, nb_actions)
        self.fc_embed = nn.Linear(nb_pstate, 32)
        self.fc11 = nn.Linear(256 + 32 + nb_actions, 128)
        self.fc21 = nn.Linear(128, 32)
        self.fc31 = nn.Linear(32, nb_actions)
        self.apply(weights_init_)

    def forward(self, inp):
        istate, pstate, a = inp
        x1 = self._features(istate)
        x2 = _lin(self.fc_embed, pstate, relu=True)
        (q1,), (q2,) = _head([x1, x2, a], [(self.fc1, self.fc2, [self.fc3]), (self.fc11, self.fc21, [self.fc31])])
        return q1, q2


class GaussianPolicy(_ConvStack):
    """Tanh-Gaussian CNN actor (got_sac_network.py:258-327)."""

    def __init__(self, nb_actions, nb_pstate, action_space=None):
        super().__init__()
        self._make_convs()
        self.fc_embed = nn.Linear(nb_pstate, 32)
        self.fc1 = nn.Linear(256 + 32, 128)
        self.fc2 = nn.Linear(128, 32)
        self.mean_linear = nn.Linear(32, nb_actions)
        self.log_std_linear = nn.Linear(32, nb_actions)
        self.apply(weights_init_)
        self.action_scale, self.action_bias = _action_affine(action_space)

    def _head_outputs(self, inp):
        istate, pstate = inp
        xs = [self._features(istate), _lin(self.fc_embed, pstate)]                    # no activation on the goal (:299)
        ((mean, log_std_raw),) = _head(xs, [(self.fc1, self.fc2, [self.mean_linear, self.log_std_linear])])
        return mean, log_std_raw

    def forward(self, inp):
        mean, log_std_raw = self._head_outputs(inp)
        return mean, torch.clamp(log_std_raw, min=LOG_SIG_MIN, max=LOG_SIG_MAX)

    def sample(self, inp):
        mean, log_std_raw = self._head_outputs(inp)
        return _S._tanh_gaussian(self, mean, log_std_raw)

    def to(self, device):
        self.action_scale = self.action_scale.to(device)
        self.action_bias = self.action_bias.to(device)
        return super().to(device)
