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
from .sac_networks import LOG_SIG_MAX, LOG_SIG_MIN, epsilon, weights_init_, _lin, _action_affine


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
        x = torch.cat([x1, x2, a], dim=1)
        q1 = _lin(self.fc3, _lin(self.fc2, _lin(self.fc1, x, True), True))
        q2 = _lin(self.fc31, _lin(self.fc21, _lin(self.fc11, x, True), True))
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

    def forward(self, inp):
        istate, pstate = inp
        x = torch.cat([self._features(istate), _lin(self.fc_embed, pstate)], dim=1)   # no activation on the goal (:299)
        x = _lin(self.fc2, _lin(self.fc1, x, True), True)
        mean = _lin(self.mean_linear, x)
        log_std = torch.clamp(_lin(self.log_std_linear, x), min=LOG_SIG_MIN, max=LOG_SIG_MAX)
        return mean, log_std

    def sample(self, inp):
        mean, log_std = self.forward(inp)
        std = log_std.exp()
        normal = Normal(mean, std, validate_args=False)
        x_t = normal.rsample()
        y_t = torch.tanh(x_t)
        action = y_t * self.action_scale + self.action_bias
        log_prob = normal.log_prob(x_t) - torch.log(self.action_scale * (1 - y_t.pow(2)) + epsilon)
        log_prob = log_prob.sum(1, keepdim=True)
        mean = torch.tanh(mean) * self.action_scale + self.action_bias
        return action, log_prob, mean

    def to(self, device):
        self.action_scale = self.action_scale.to(device)
        self.action_bias = self.action_bias.to(device)
        return super().to(device)
