"""Adam over the flat parameter / gradient buffers (`raindrop_amd.dp.FlatGradAllReduce`).

`code/Raindrop.py:256` uses `torch.optim.Adam(model.parameters(), lr=1e-4)`; that keeps working
with this model unchanged.  `FlatAdam` is the fast path: because the live parameters and their
gradients already live in two flat buffers, the whole update is ONE elementwise HIP kernel
(rd_adam_step) instead of a 35-tensor multi-tensor launch.  Same formula as torch (no amsgrad).
"""
import ctypes

import torch

from . import _lib, ops


class FlatAdam:
    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if flat_param.grad is None:
            raise ValueError("flat_param.grad must be the flat gradient buffer")
        self.param = flat_param
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat_param.data)
        self.exp_avg_sq = torch.zeros_like(flat_param.data)
        self.t = 0
        self.step_cell = None                # device step state of the captured form (step_captured)
        self._cell_stale = False

    def hyper(self):
        """the values a captured launch holds as CONSTANTS (TrainStep.run_full captures again when they change): betas and eps.
        lr and weight_decay live in the device step cell (cell_hyper): changing them is a small copy, not a new capture."""
        return (float(self.betas[0]), float(self.betas[1]), float(self.eps))

    def cell_hyper(self):
        return (float(self.lr), float(self.weight_decay))

    def step(self):
        self.t += 1
        self._cell_stale = True              # a captured step that follows re-syncs the device step state (TrainStep.run_full)
        p, g = self.param.data, self.param.grad
        _lib.call("rd_adam_step", p.numel(), ops._ptr(p), ops._ptr(g), ops._ptr(self.exp_avg), ops._ptr(self.exp_avg_sq),
                  float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                  self.t, ops._stream())

    def step_captured(self, advance=True):
        """The update as launches a hipGraph can hold (rd_adam_step_dev: step count, beta^t, lr and weight decay live on the device).
        advance=True: a one-thread launch first moves the device state to this step (rd_adam_state_advance); advance=False: the
        caller's step already did (raindrop_amd.step.TrainStep.capture_full registers the cell -- `register_cell` -- and its first
        launch advances it: no extra launch).  Every replay is one step -- the owner of the graph keeps `self.t` in step
        (`note_replay`) and pushes a changed lr / weight decay with `sync_cell_hyper` (no new capture)."""
        self.sync_step_cell(create_only=True)
        p, g = self.param.data, self.param.grad
        if advance:
            _lib.call("rd_adam_state_advance", ops._ptr(self.step_cell), float(self.betas[0]), float(self.betas[1]), ops._stream())
        _lib.call("rd_adam_step_dev", p.numel(), ops._ptr(p), ops._ptr(g), ops._ptr(self.exp_avg), ops._ptr(self.exp_avg_sq),
                  float(self.betas[0]), float(self.betas[1]), float(self.eps), ops._ptr(self.step_cell), ops._stream())

    def register_cell(self, on=True):
        """(Un)register the device step state with this host thread's next rd_step_begin launches (include/raindrop_hip.h
        rd_set_adam_state): the step's first launch then advances it."""
        self.sync_step_cell(create_only=True)
        _lib.call("rd_set_adam_state", ops._ptr(self.step_cell) if on else None, float(self.betas[0]), float(self.betas[1]))

    def sync_step_cell(self, create_only=False):
        """Make the device step state agree with `self.t`, `self.lr`, `self.weight_decay` (before capturing, after load_state_dict,
        after eager steps).  Layout (include/raindrop_hip.h rd_adam_step_dev): {t, beta1^t, beta2^t, lr, 0, wd, 0, 0} with t = steps TAKEN (the step's
        advance launch moves it to the step being applied)."""
        fresh = getattr(self, "step_cell", None) is None
        if fresh:
            self.step_cell = torch.zeros((8,), dtype=torch.float64, device=self.param.device)
        if fresh or not create_only:
            # beta^t of the betas AS THE KERNELS SEE THEM (float arguments widened to double: rd_adam_step's pow() and the device's
            # running product both start from those) -- 0.999 as a double instead of 0.999f moves 1 - beta2^t by 1e-5 relative
            t = float(self.t)
            b1, b2 = (ctypes.c_float(float(b)).value for b in self.betas)
            self.step_cell.copy_(torch.tensor([t, b1 ** t, b2 ** t, float(self.lr), 0.0, float(self.weight_decay), 0.0, 0.0],
                                              dtype=torch.float64))
            self._cell_hyper = self.cell_hyper()

    def sync_cell_hyper(self):
        """lr / weight decay changed on the host (ReduceLROnPlateau, a warm-up or cosine schedule): two 8-byte cells, stream-ordered
        with the replays around it."""
        if self.step_cell is not None and getattr(self, "_cell_hyper", None) != self.cell_hyper():
            self.step_cell[3:6:2] = torch.tensor([float(self.lr), float(self.weight_decay)], dtype=torch.float64,
                                                 device=self.step_cell.device)
            self._cell_hyper = self.cell_hyper()

    def device_steps(self):
        """steps taken according to the device state"""
        return int(float(self.step_cell[0]))

    def note_replay(self):
        self.t += 1

    def zero_grad(self, set_to_none=False):
        self.param.grad.zero_()

    def state_dict(self):
        return dict(t=self.t, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr, betas=self.betas,
                    eps=self.eps, weight_decay=self.weight_decay)

    def load_state_dict(self, sd):
        """In place (the moment buffers keep their addresses: a captured step stays valid); the device step state follows at the
        next captured step."""
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.t = int(sd["t"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
        self._cell_stale = True
