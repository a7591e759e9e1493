"""Minimal consumer of gathered (state, pi, z) rows for BASELINE config 5 (self-play + concurrent training).

The reference's trainer (train.py) is OUT OF SCOPE for the rollout path: it is a plain
PyTorch consumer with no custom ops. Config 5 still needs something to consume the replay buffer on
GPU0, so this restates just the update of train.py:163-187 -- AMP forward, ``mse(value, z) +
(-mean sum(smooth(pi) * logp))`` with label smoothing 0.05, grad-norm clip 5.0, the Adam optimiser of
net.py:121-127 -- without the KL-adaptive learning rate, rollback guards and checkpoint rotation.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class Trainer:
    def __init__(self, policy_value_net, label_smoothing: float = 0.05, clip: float = 5.0, lr: float = 1e-3, amp_dtype: str = "fp16"):
        """``amp_dtype``: "fp16" = autocast + GradScaler as train.py:163-187 (the scaler's step() reads ``found_inf`` on the
        host: one sync per update); "bf16" = bf16 autocast without a scaler, no host sync at all (what bench.py's concurrent
        config-5 trainer uses, so that ``step(sync=False)`` really never waits)."""
        self.pvn = policy_value_net
        self.net = policy_value_net.policy_value_net
        self.opt = policy_value_net.optimizer
        self.eps = label_smoothing
        self.clip = clip
        for g in self.opt.param_groups:
            g["lr"] = lr
        if amp_dtype not in ("fp16", "bf16"):
            raise ValueError("amp_dtype must be 'fp16' or 'bf16'")
        self.use_amp = next(self.net.parameters()).is_cuda
        self.amp_dtype = torch.float16 if amp_dtype == "fp16" else torch.bfloat16
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.use_amp and amp_dtype == "fp16")
        self.steps = 0

    def step(self, states: torch.Tensor, pi: torch.Tensor, z: torch.Tensor, sync: bool = True) -> dict:
        self.pvn._require_current_fp32("Trainer.step")   # a rank that received only the inference copy holds OLD fp32 weights
        self.net.train()
        dev = next(self.net.parameters()).device
        states, pi, z = states.to(dev).float(), pi.to(dev).float(), z.to(dev).float()
        if sync:
            sums = pi.sum(dim=1)
            if not ((sums > 0.99) & (sums < 1.01)).all():  # train.py:134-136
                raise ValueError("mcts_probs rows must sum to 1 (+-0.01)")
        self.opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.use_amp):
            log_act_probs, value = self.net(states)
            value_loss = F.mse_loss(value.flatten().float(), z)
            target = (1 - self.eps) * pi + self.eps / pi.size(1) if self.eps > 0 else pi
            policy_loss = -torch.mean(torch.sum(target * log_act_probs.float(), dim=1))
            loss = value_loss + policy_loss
        self.scaler.scale(loss).backward()
        self.scaler.unscale_(self.opt)
        torch.nn.utils.clip_grad_norm_(self.net.parameters(), self.clip)
        self.scaler.step(self.opt)
        self.scaler.update()
        self.steps += 1
        with torch.no_grad():
            entropy = -torch.mean(torch.sum(torch.exp(log_act_probs.float()) * log_act_probs.float(), dim=1))
        self.pvn.invalidate_inference_copy()  # stale now (and captured hipGraphs with it); rebuilt on the next evaluation
        out = {"loss": loss.detach(), "policy_loss": policy_loss.detach(), "value_loss": value_loss.detach(), "entropy": entropy}
        return {k: float(v) for k, v in out.items()} if sync else out  # sync=False: no host wait (concurrent self-play)
