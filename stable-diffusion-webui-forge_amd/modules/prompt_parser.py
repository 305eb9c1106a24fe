"""The conditioning object shapes of modules/prompt_parser.py that the denoiser consumes (:271-365), for prompts
without scheduling: `reconstruct_cond_batch` / `reconstruct_multicond_batch` accept either ready tensors
(`[B,T,D]` or `DictWithShape{crossattn, vector}`) or the reference's scheduled lists, whose schedule is resolved per
step.  Text encoding itself is out of scope; BASELINE configs use synthetic cond tensors."""
import torch


class DictWithShape(dict):
    """prompt_parser.py:271-291"""

    def __init__(self, x, shape=None):
        super().__init__()
        self.update(x)

    @property
    def shape(self):
        return self["crossattn"].shape

    def to(self, *args, **kwargs):
        for k in self.keys():
            if isinstance(self[k], torch.Tensor):
                self[k] = self[k].to(*args, **kwargs)
        return self

    def advanced_indexing(self, item):
        return DictWithShape({k: v[item] for k, v in self.items()})


def _as_cond(c):
    if isinstance(c, dict) and not isinstance(c, DictWithShape):
        return DictWithShape(c)
    return c


def reconstruct_cond_batch(c, current_step):
    """:294-318 -- for plain (unscheduled) conds the tensor itself."""
    if isinstance(c, (torch.Tensor, dict)):
        return _as_cond(c)
    raise NotImplementedError("scheduled prompt lists need the text encoders (out of scope); pass cond tensors")


def reconstruct_multicond_batch(c, current_step):
    """:337-365 -- returns (conds_list, cond): one (index, weight=1.0) entry per image for plain prompts."""
    cond = reconstruct_cond_batch(c, current_step)
    b = cond["crossattn"].shape[0] if isinstance(cond, dict) else cond.shape[0]
    return [[(i, 1.0)] for i in range(b)], cond
