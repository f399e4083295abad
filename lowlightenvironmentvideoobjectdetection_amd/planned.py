"""Mixin that turns a parameter-container nn.Module into a HIP-plan-backed model."""
import torch

from ._lib import LlvError


class PlannedModule(torch.nn.Module):
    """Plans are cached per input geometry and dropped whenever parameters may have changed
    (load_state_dict, train(), .to(), explicit invalidate_plans())."""

    def __init__(self):
        super().__init__()
        self._plans = {}

    def invalidate_plans(self):
        """Drop the cached plans (packed weights, launch lists, graphs) of this module AND of every module below it:
        nn.Module.load_state_dict on a parent never calls its children's overrides, so a container (TwoStageDetector, SELSA,
        a video model) that is re-loaded, re-initialised or stepped by an optimizer has to reach its backbone / neck / heads
        itself, or they keep serving convolutions packed from the old weights."""
        for m in self.modules():
            if isinstance(m, PlannedModule):
                m._plans = {}

    def load_state_dict(self, *a, **k):
        self.invalidate_plans()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        """Switching between training and evaluation drops the plans: parameters are updated in place while training
        (optimizer.step() outside forward_train), and the evaluation plans fold them at build time."""
        if bool(mode) != self.training:
            self.invalidate_plans()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._plans = {}
        return super()._apply(fn, *a, **k)

    def _plan(self, key, make):
        if key not in self._plans:
            self._plans[key] = make()
        return self._plans[key]

    @staticmethod
    def _check_input(x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise TypeError('expected a 4-D NCHW tensor')
        if not x.is_cuda:
            raise LlvError('this model runs on an MI355X only (HIP kernels through libllv.so); for a CPU result use '
                           'the oracle/ restatement in tests, it is not a fallback')
