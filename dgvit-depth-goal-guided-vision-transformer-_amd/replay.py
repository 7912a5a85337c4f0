"""Device-resident replay storage and sampling.

The reference keeps transitions in a host-side ``cpprb.PrioritizedReplayBuffer`` used as a plain uniform sampler
(priorities are never updated, DRL.py:80-100, 365-368) and, every ``learn()``, converts the sampled numpy batch to
tensors and copies it to the device synchronously from pageable memory (DRL.py:375-386): 2 x (B,128,160) fp32 =
84 MB at B=512.  An MI355X has 288 GB of HBM: 100k transitions of two 128x160 fp32 frames are 16 GB, so the ring
lives on the device, ``store_transition`` becomes one small H2D copy per environment step, and ``sample`` is an
index draw plus one HBM-bound gather kernel per field -- the encoder never waits for PCIe.

Field names follow the reference's buffer (DRL.py:80-100): obs, pobs, act, rew, next_obs, next_pobs, done.
"""
import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib


def _al4(n: int) -> int:
    return (n + 3) & ~3


class DeviceReplayBuffer:
    def __init__(self, size: int, obs_shape: Tuple[int, int] = (128, 160), pstate_dim: int = 2, act_dim: int = 2,
                 device="cuda", seed: Optional[int] = None):
        self.size, self.device = int(size), torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DgvitError("DeviceReplayBuffer lives in HBM; pass a ROCm device")
        self.fields = {"obs": int(np.prod(obs_shape)), "pobs": pstate_dim, "act": act_dim, "rew": 1,
                       "next_obs": int(np.prod(obs_shape)), "next_pobs": pstate_dim, "done": 1}
        self.shapes = {"obs": tuple(obs_shape), "next_obs": tuple(obs_shape), "pobs": (pstate_dim,), "next_pobs": (pstate_dim,),
                       "act": (act_dim,), "rew": (1,), "done": (1,)}
        # every field is a (size, row) fp32 matrix whose row length is padded to a multiple of 4 floats (float4 gather)
        self.rows = {k: _al4(n) for k, n in self.fields.items()}
        self.store = {k: torch.zeros(self.size, r, dtype=torch.float32, device=self.device) for k, r in self.rows.items()}
        self.next_index, self.stored = 0, 0
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def get_stored_size(self) -> int:
        return self.stored

    def _host_rows(self, kw, n):
        """(n, row) fp32 host matrix holding the fields of n transitions side by side (offsets in self._off)."""
        if not hasattr(self, "_off"):
            self._off, o = {}, 0
            for k, r in self.rows.items():
                self._off[k] = o
                o += r
            self._row = o
        host = np.zeros((n, self._row), dtype=np.float32)
        for k, cnt in self.fields.items():
            if k not in kw:
                raise KeyError(f"missing field {k}")
            v = kw[k].detach().cpu().numpy() if torch.is_tensor(kw[k]) else np.asarray(kw[k], dtype=np.float32)
            v = np.asarray(v, dtype=np.float32).reshape(n, -1)
            if v.shape[1] != cnt:
                raise ValueError(f"{k}: expected {cnt} values per transition, got {v.shape[1]}")
            host[:, self._off[k]:self._off[k] + cnt] = v
        return host

    def _store_rows(self, host) -> None:
        """ONE host->device copy of the assembled rows (pinned staging), then device-side copies into the field matrices
        (ring wrap = two slices).  The reference converts and copies every field of every sampled batch instead (DRL.py:379-386)."""
        n = host.shape[0]
        if n > self.size:
            host, n = host[-self.size:], self.size
        stage = torch.from_numpy(host).pin_memory()
        dev = stage.to(self.device, non_blocking=True)
        i = self.next_index
        first = min(n, self.size - i)
        for k, cnt in self.fields.items():
            o = self._off[k]
            self.store[k][i:i + first, :cnt].copy_(dev[:first, o:o + cnt])
            if first < n:
                self.store[k][:n - first, :cnt].copy_(dev[first:, o:o + cnt])
        self._last_stage = (stage, dev)          # keep the pinned buffer alive until the async copy has certainly been issued
        self.next_index = (i + n) % self.size
        self.stored = min(self.stored + n, self.size)

    def add(self, **kw) -> None:
        """One transition (numpy arrays / scalars / tensors), same keywords as replay_buffer.add in DRL.py:439-448:
        one staged host->device copy for the whole transition."""
        self._store_rows(self._host_rows(kw, 1))

    def add_batch(self, **kw) -> None:
        """Many transitions at once (first axis = transitions), e.g. the expert demonstrations of DRL.py:469-478: assembled on
        the host with numpy slicing, one host->device copy."""
        self._store_rows(self._host_rows(kw, len(kw["obs"])))

    def sample_indices(self, batch_size: int) -> torch.Tensor:
        if self.stored == 0:
            raise RuntimeError("cannot sample from an empty buffer")
        return torch.randint(0, self.stored, (batch_size,), device=self.device, dtype=torch.int64, generator=self.gen)

    def sample(self, batch_size: int, indices: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Uniform sample with replacement -> dict of DEVICE tensors shaped like the reference's batch
        (obs (B,H,W), pobs (B,2), act (B,2), rew (B,1), ...), ready for the networks: no host round trip."""
        lib = _lib.load()
        idx = self.sample_indices(batch_size) if indices is None else indices.to(self.device, torch.int64).contiguous()
        B = idx.numel()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        out = {}
        with torch.cuda.device(self.device):
            for k, n in self.fields.items():
                r = self.rows[k]
                buf = torch.empty(B, r, dtype=torch.float32, device=self.device)
                rc = lib.dgvit_gather_rows(ctypes.c_void_p(self.store[k].data_ptr()), ctypes.c_void_p(idx.data_ptr()),
                                           ctypes.c_void_p(buf.data_ptr()), B, r, self.size, st)
                _lib.check(rc, "dgvit_gather_rows")
                out[k] = buf[:, :n].reshape(B, *self.shapes[k])
        out["indexes"] = idx
        return out
