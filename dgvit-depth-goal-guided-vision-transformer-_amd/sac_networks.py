"""MI355X-native twins of the GoT networks in the reference's ``got_sac_network.py``:
``GoTPolicy`` (:172-256), ``GoTQNetwork`` (:75-123), ``DeterministicGoTPolicy`` (:389-449).

Constructor signatures, attribute names (``trans``, ``fc_embed``, ``fc1`` ... used by DRL.py:107-111,145-148 to pick
optimiser sub-sets) and ``state_dict`` keys match the reference.  The encoder and every Linear run in
libdgvit_hip.so; sampling (Normal / tanh / clamp on (B, 2) tensors) stays in PyTorch.

``image_size`` / ``patch_size`` are extra keyword arguments (default = the reference's hard-wired 128x160 @ 16x20).
"""
import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from . import functional as F_
from .goalformer import GoT

LOG_SIG_MAX = 2
LOG_SIG_MIN = -20
epsilon = 1e-6


def set_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def weights_init_(m):
    """Xavier-uniform (gain 1) on every nn.Linear weight; biases keep PyTorch's default (got_sac_network.py:30-33)."""
    if isinstance(m, nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight, gain=1)


def _lin(layer, x, relu=False):
    return F_.linear(x, layer.weight, layer.bias, relu=relu)


def _head(xs, towers):
    """relu(fc1) -> relu(fc2) -> third layer(s) on cat(xs): ONE fused HIP launch each way when the widths fit the fused head
    kernel (every network of the reference does), else the same arithmetic as a chain of MFMA GEMM launches.
    Returns [[y of third layer j of tower t]]."""
    if xs[0].shape[0] > 0 and F_.mlp_head_supported(xs, towers):
        return F_.mlp_head(xs, towers)
    x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
    out = []
    for l1, l2, l3s in towers:
        h = _lin(l2, _lin(l1, x, True), True)
        out.append([_lin(l3, h) for l3 in l3s])
    return out


def _standard_normal(like):
    """The N(0, 1) draw inside Normal.rsample (got_sac_network.py:242), made on the device.  Tests replace this function to
    inject the reference run's own draw."""
    return torch.randn_like(like)


def _tanh_gaussian(module, mean, log_std_raw):
    """sample() of got_sac_network.py:238-251 / :310-321 from the head outputs: (action, log_prob, tanh(mean) * scale + bias)."""
    if mean.shape[0] == 0:       # empty batch: the reference's torch ops accept it
        log_std = torch.clamp(log_std_raw, min=LOG_SIG_MIN, max=LOG_SIG_MAX)
        normal = Normal(mean, log_std.exp(), validate_args=False)
        x_t = normal.rsample()
        y_t = torch.tanh(x_t)
        log_prob = (normal.log_prob(x_t) - torch.log(module.action_scale * (1 - y_t.pow(2)) + epsilon)).sum(1, keepdim=True)
        return y_t * module.action_scale + module.action_bias, log_prob, torch.tanh(mean) * module.action_scale + module.action_bias
    scale, bias = module.action_scale, module.action_bias
    if scale.device != mean.device:          # module moved with .cuda() instead of the reference's .to(): follow it once
        module.action_scale, module.action_bias = scale, bias = scale.to(mean.device), bias.to(mean.device)
    return F_.tanh_gaussian_sample(mean, log_std_raw, _standard_normal(mean), scale, bias, LOG_SIG_MIN, LOG_SIG_MAX)


def _action_affine(action_space):
    if action_space is None:
        return torch.tensor(1.), torch.tensor(0.)
    return (torch.FloatTensor((action_space.high - action_space.low) / 2.),
            torch.FloatTensor((action_space.high + action_space.low) / 2.))


def _encoder(dim, depth, heads, image_size, patch_size):
    return GoT(image_size=image_size, patch_size=patch_size, num_classes=2, dim=dim, depth=depth, heads=heads, mlp_dim=2048,
               channels=1)


class GoTQNetwork(nn.Module):
    """Twin-Q critic on GoT features: forward([istate, pstate, a]) -> (q1, q2), each (B, nb_actions)."""

    def __init__(self, nb_actions, nb_pstate, block, head, l_f_size, image_size=(128, 160), patch_size=(16, 20)):
        super().__init__()
        self.trans = _encoder(l_f_size, block, head, image_size, patch_size)
        # dead parameters of the reference (got_sac_network.py:90-92): kept so checkpoints load strictly
        self.conv1 = nn.Conv2d(4, 16, 5, stride=2)
        self.conv2 = nn.Conv2d(16, 64, 5, stride=2)
        self.conv3 = nn.Conv2d(64, 256, 5, stride=2)
        self.avg = nn.AdaptiveAvgPool2d(output_size=(1, 1))
        self.fc1 = nn.Linear(l_f_size + nb_actions, 128)
        self.fc2 = nn.Linear(128, 32)
        self.fc3 = nn.Linear(32, nb_actions)
        self.fc_embed = nn.Linear(nb_pstate, l_f_size)
        self.fc11 = nn.Linear(l_f_size + nb_actions, 128)
        self.fc21 = nn.Linear(128, 32)
        self.fc31 = nn.Linear(32, nb_actions)
        self.apply(weights_init_)

    def forward(self, inp):
        istate, pstate, a = inp
        goal = _lin(self.fc_embed, pstate, relu=True)          # ReLU on the goal embedding (:111)
        feat = self.trans(istate, goal)
        (q1,), (q2,) = _head([feat.view(feat.size(0), -1), a], [(self.fc1, self.fc2, [self.fc3]), (self.fc11, self.fc21, [self.fc31])])
        return q1, q2


class GoTPolicy(nn.Module):
    """Tanh-Gaussian actor on GoT features."""

    def __init__(self, nb_actions, nb_pstate, block, head, l_f_size, action_space=None, image_size=(128, 160),
                 patch_size=(16, 20)):
        super().__init__()
        self.trans = _encoder(l_f_size, block, head, image_size, patch_size)
        self.fc_embed = nn.Linear(nb_pstate, l_f_size)
        self.fc1 = nn.Linear(l_f_size, 128)
        self.fc2 = nn.Linear(128, 128)
        self.mean_linear = nn.Linear(128, nb_actions)
        self.log_std_linear = nn.Linear(128, nb_actions)
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.apply(weights_init_)
        self.action_scale, self.action_bias = _action_affine(action_space)

    def choose_action(self, istate, pstate, evaluate=False):
        dev = self.fc1.weight.device
        if istate.ndim < 4:
            istate = torch.FloatTensor(istate).float().permute(2, 0, 1)      # (H, W, 1) -> (1, H, W)
            pstate = torch.FloatTensor(pstate).float().unsqueeze(0)
        else:
            istate = torch.FloatTensor(istate).float().permute(0, 3, 1, 2)
            pstate = torch.FloatTensor(pstate).float()
        istate, pstate = istate.to(dev), pstate.to(dev)
        if evaluate is False:
            action, _, _ = self.sample([istate, pstate])
        else:
            _, _, action = self.sample([istate, pstate])
        return action.detach().squeeze(0).cpu().numpy()

    def _head_outputs(self, inp):
        istate, pstate = inp
        goal = _lin(self.fc_embed, pstate)                      # no activation (:226)
        feat = self.trans(istate, goal)
        ((mean, log_std_raw),) = _head([feat], [(self.fc1, self.fc2, [self.mean_linear, self.log_std_linear])])
        return mean, log_std_raw

    def forward(self, inp):
        mean, log_std_raw = self._head_outputs(inp)
        return mean, torch.clamp(log_std_raw, min=LOG_SIG_MIN, max=LOG_SIG_MAX)

    def sample(self, inp):
        """(action, log_prob, tanh(mean)) of got_sac_network.py:238-251: clamp, exp, rsample, tanh and the tanh-corrected Gaussian
        log-density as ONE HIP launch behind the head (no Normal object: its argument check alone is a host sync per call)."""
        mean, log_std_raw = self._head_outputs(inp)
        return _tanh_gaussian(self, mean, log_std_raw)

    def to(self, device):
        self.action_scale = self.action_scale.to(device)
        self.action_bias = self.action_bias.to(device)
        return super().to(device)


class DeterministicGoTPolicy(nn.Module):
    def __init__(self, nb_actions, nb_pstate, block, head, l_f_size, action_space=None, image_size=(128, 160),
                 patch_size=(16, 20)):
        super().__init__()
        self.trans = _encoder(l_f_size, block, head, image_size, patch_size)
        self.fc_embed = nn.Linear(nb_pstate, l_f_size)
        self.fc1 = nn.Linear(l_f_size, 128)
        self.fc2 = nn.Linear(128, 32)
        self.noise = torch.Tensor(nb_actions)
        self.apply(weights_init_)
        # created after the Xavier pass, as in the reference (:410-413): default nn.Linear init
        self.mean_linear = nn.Linear(32, nb_actions)
        self.log_std_linear = nn.Linear(32, nb_actions)
        self.action_scale, self.action_bias = _action_affine(action_space)

    def forward(self, inp):
        istate, pstate = inp
        goal = _lin(self.fc_embed, pstate)
        feat = self.trans(istate, goal)
        ((mean,),) = _head([feat.view(feat.size(0), -1)], [(self.fc1, self.fc2, [self.mean_linear])])
        return torch.tanh(mean) * self.action_scale + self.action_bias

    def sample(self, inp):
        mean = self.forward(inp)
        noise = self.noise.normal_(0., std=0.1).clamp(-0.25, 0.25)
        return mean + noise, torch.tensor(0.), mean

    def to(self, device):
        self.action_scale = self.action_scale.to(device)
        self.action_bias = self.action_bias.to(device)
        self.noise = self.noise.to(device)
        return super().to(device)
