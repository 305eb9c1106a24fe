"""The conditioning objects of modules/prompt_parser.py that the denoiser consumes (:129-147, :233-242, :271-365):
`ScheduledPromptConditioning` lists (prompt editing: the cond changes at given steps), `MulticondLearnedConditioning` (AND-composed
prompts: several weighted conds per image) and their per-step resolution `reconstruct_cond_batch` / `reconstruct_multicond_batch` /
`stack_conds`.  Ready tensors (`[B,T,D]` or `DictWithShape{crossattn, vector}`) are accepted wherever a schedule is, standing for plain
prompts.  `get_multicond_prompt_list` (the AND / `:weight` splitting, :205-230) and `get_learned_conditioning` / `get_multicond_learned_conditioning`
(:150-202, :245-268) and the `[from:to:when]` / `[a|b]` prompt-editing grammar (:7-127; a lark Earley grammar in the reference, an ordered-choice
recursive-descent parser here, checked against the reference's own doctests) are host text processing that feeds them."""
import re
from collections import namedtuple

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


ScheduledPromptConditioning = namedtuple("ScheduledPromptConditioning", ["end_at_step", "cond"])  # prompt_parser.py:129


class ComposableScheduledPromptConditioning:
    """One AND-part of a prompt: its schedule and its weight (:233-236)."""

    def __init__(self, schedules, weight=1.0):
        self.schedules = schedules
        self.weight = weight


class MulticondLearnedConditioning:
    """:239-242; `shape` lets the object stand where a tensor is expected (DDIM / PLMS look at it)."""

    def __init__(self, shape, batch):
        self.shape = shape
        self.batch = batch


class SdConditioning(list):
    """:132-147: prompts (or token batches) + the side information SDXL's conditioner needs."""

    def __init__(self, prompts, is_negative_prompt=False, width=None, height=None, copy_from=None, distilled_cfg_scale=None):
        super().__init__()
        self.extend(prompts)
        if copy_from is None:
            copy_from = prompts
        self.is_negative_prompt = is_negative_prompt or getattr(copy_from, "is_negative_prompt", False)
        self.width = width or getattr(copy_from, "width", None)
        self.height = height or getattr(copy_from, "height", None)
        self.distilled_cfg_scale = distilled_cfg_scale or getattr(copy_from, "distilled_cfg_scale", None)


# ---- prompt editing: "[from:to:when]", "[to:when]", "[from::when]", "[a|b|c]" (prompt_parser.py:7-127) ---------------------------------------
# The reference parses this with a lark Earley grammar (:18-29).  The language is small enough for ordered-choice recursive descent: inside a
# prompt, "[" opens (in this order) a scheduled, an alternate or an emphasised group, "(" an emphasised group; a bracket that opens none of them
# ends the enclosing prompt, and at the top level the six punctuation characters "[]():" (not "|") may also stand alone as literal text.
class _NoParse(Exception):
    pass


_PLAIN = re.compile(r"(?:[^\\\[\]():|]|\\.)+", re.S)
_WS = re.compile(r"\s+")
_NUMBER = re.compile(r"[+-]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?")


def _parse_prompt(s, i):
    """prompt: (emphasized | scheduled | alternate | plain | WHITESPACE)*  ->  (nodes, next index)."""
    nodes = []
    while i < len(s):
        ch = s[i]
        m = _PLAIN.match(s, i)
        if m:
            nodes.append(m.group(0))
            i = m.end()
        elif ch == "[":
            for rule in (_parse_scheduled, _parse_alternate, _parse_square):
                try:
                    node, i = rule(s, i)
                    break
                except _NoParse:
                    continue
            else:
                return nodes, i
            nodes.append(node)
        elif ch == "(":
            try:
                node, i = _parse_round(s, i)
            except _NoParse:
                return nodes, i
            nodes.append(node)
        else:
            return nodes, i
    return nodes, i


def _expect(s, i, ch):
    if i >= len(s) or s[i] != ch:
        raise _NoParse
    return i + 1


def _parse_scheduled(s, i):
    """"[" [prompt ":"] prompt ":" [WS] NUMBER [WS] "]" """
    i = _expect(s, i, "[")
    first, i = _parse_prompt(s, i)
    i = _expect(s, i, ":")

    def tail(j):
        m = _WS.match(s, j)
        j = m.end() if m else j
        num = _NUMBER.match(s, j)
        if not num:
            raise _NoParse
        j = num.end()
        m = _WS.match(s, j)
        j = m.end() if m else j
        return num.group(0), _expect(s, j, "]")
    try:  # two-part form first: "[" prompt ":" prompt ":" NUMBER "]"
        second, j = _parse_prompt(s, i)
        j = _expect(s, j, ":")
        when, j = tail(j)
        return ("scheduled", first, second, when), j
    except _NoParse:
        when, j = tail(i)
        return ("scheduled", None, first, when), j


def _parse_alternate(s, i):
    """"[" prompt ("|" [prompt])+ "]" """
    i = _expect(s, i, "[")
    first, i = _parse_prompt(s, i)
    options = [first]
    i = _expect(s, i, "|")
    while True:
        opt, i = _parse_prompt(s, i)
        options.append(opt)
        if i < len(s) and s[i] == "|":
            i += 1
            continue
        return ("alternate", options), _expect(s, i, "]")


def _parse_square(s, i):
    i = _expect(s, i, "[")
    inner, i = _parse_prompt(s, i)
    return ("group", "[", inner, "]"), _expect(s, i, "]")


def _parse_round(s, i):
    """"(" prompt ")" | "(" prompt ":" prompt ")" """
    i = _expect(s, i, "(")
    inner, i = _parse_prompt(s, i)
    if i < len(s) and s[i] == ":":
        second, j = _parse_prompt(s, i + 1)
        return ("group", "(", inner + [":"] + second, ")"), _expect(s, j, ")")
    return ("group", "(", inner, ")"), _expect(s, i, ")")


def _parse_schedule_tree(s):
    """start: (prompt | /[][():]/+)*"""
    nodes, i = [], 0
    while i < len(s):
        part, j = _parse_prompt(s, i)
        nodes += part
        if j < len(s):
            if s[j] not in "[]():":
                raise _NoParse  # e.g. a top-level "|": no parse, the prompt is used as is (:119-124)
            nodes.append(s[j])
            j += 1
        i = j
    return nodes


def get_learned_conditioning_prompt_schedules(prompts, base_steps, hires_steps=None, use_old_scheduling=False):
    """:31-127 -> per prompt [[end_at_step, text], ...].  `when` with a decimal point is a fraction of the steps, otherwise a step number;
    in a hires pass (hires_steps given) integers count on from base_steps and fractions from 1.0 (:49-57)."""
    if hires_steps is None or use_old_scheduling:
        int_offset, flt_offset, steps = 0, 0, base_steps
    else:
        int_offset, flt_offset, steps = base_steps, 1.0, hires_steps

    def when_of(token):
        v = float(token)
        if use_old_scheduling:
            v = v * steps if v < 1 else v
        elif "." in token:
            v = (v - flt_offset) * steps
        else:
            v = v - int_offset
        return min(steps, int(v))

    def collect(nodes, out):
        for n in nodes:
            if isinstance(n, tuple):
                if n[0] == "scheduled":
                    w = when_of(n[3])
                    if w >= 1:
                        out.add(w)
                    collect(n[1] or [], out)
                    collect(n[2], out)
                elif n[0] == "alternate":
                    out.update(range(1, steps + 1))
                    for o in n[1]:
                        collect(o, out)
                else:
                    collect(n[2], out)

    def render(nodes, step):
        parts = []
        for n in nodes:
            if isinstance(n, str):
                parts.append(n)
            elif n[0] == "scheduled":
                parts.append(render(n[1] or [], step) if step <= when_of(n[3]) else render(n[2], step))
            elif n[0] == "alternate":
                parts.append(render(n[1][(step - 1) % len(n[1])], step))
            else:
                parts.append(n[1] + render(n[2], step) + n[3])
        return "".join(parts)

    def schedule(prompt):
        try:
            tree = _parse_schedule_tree(prompt)
        except _NoParse:
            return [[steps, prompt]]
        marks = {steps}
        collect(tree, marks)
        return [[t, render(tree, t)] for t in sorted(marks)]
    cache = {p: schedule(p) for p in set(prompts)}
    return [cache[p] for p in prompts]


_AND = re.compile(r"\bAND\b")
_WEIGHT = re.compile(r"^((?:\s|.)*?)(?:\s*:\s*([-+]?(?:\d+\.?|\d*\.\d+)))?\s*$")


def get_multicond_prompt_list(prompts):
    """:208-230: split every prompt at the AND keyword, read an optional trailing `:weight` off each part, de-duplicate identical part texts.
    -> (per prompt [(index into flat list, weight), ...], flat SdConditioning of part texts, {text: index})."""
    res_indexes, prompt_indexes = [], {}
    flat = SdConditioning(prompts)
    flat.clear()
    for prompt in prompts:
        indexes = []
        for sub in _AND.split(prompt):
            m = _WEIGHT.search(sub)
            text, weight = m.groups() if m is not None else (sub, 1.0)
            weight = float(weight) if weight is not None else 1.0
            if text not in prompt_indexes:
                prompt_indexes[text] = len(flat)
                flat.append(text)
            indexes.append((prompt_indexes[text], weight))
        res_indexes.append(indexes)
    return res_indexes, flat, prompt_indexes


def get_learned_conditioning(model, prompts, steps, hires_steps=None, use_old_scheduling=False, schedules=None):
    """:150-202 with the prompt-editing grammar factored out: `schedules[i]` = [[end_at_step, text], ...] for prompt i (what
    get_learned_conditioning_prompt_schedules returns; default: the whole prompt for all `steps`).  Every distinct text is encoded once by
    `model.get_learned_conditioning(SdConditioning(texts))` -> list of per-prompt [ScheduledPromptConditioning, ...]."""
    if schedules is None:
        schedules = get_learned_conditioning_prompt_schedules(prompts, steps, hires_steps, use_old_scheduling)
    res, cache = [], {}
    for prompt, sched in zip(prompts, schedules):
        if prompt in cache:
            res.append(cache[prompt])
            continue
        conds = model.get_learned_conditioning(SdConditioning([text for _, text in sched], copy_from=prompts))
        out = []
        for i, (end_at_step, _) in enumerate(sched):
            out.append(ScheduledPromptConditioning(end_at_step, {k: v[i] for k, v in conds.items()} if isinstance(conds, dict) else conds[i]))
        cache[prompt] = out
        res.append(out)
    return res


def get_multicond_learned_conditioning(model, prompts, steps, hires_steps=None, use_old_scheduling=False, schedules_for=None):
    """:245-268: AND-composed prompts -> MulticondLearnedConditioning.  schedules_for(flat_texts) may supply prompt-editing schedules."""
    res_indexes, flat, _ = get_multicond_prompt_list(prompts)
    learned = get_learned_conditioning(model, flat, steps, hires_steps, use_old_scheduling, None if schedules_for is None else schedules_for(flat))
    return MulticondLearnedConditioning((len(prompts),), [[ComposableScheduledPromptConditioning(learned[i], w) for i, w in indexes]
                                                         for indexes in res_indexes])


def _as_cond(c):
    if isinstance(c, dict) and not isinstance(c, DictWithShape):
        return DictWithShape(c)
    return c


def _pick(schedule, current_step):
    """index of the first entry whose end_at_step has not passed (:305-309, :346-350); 0 when all have."""
    for current, entry in enumerate(schedule):
        if current_step <= entry.end_at_step:
            return current
    return 0


_memo = {}  # id(schedule object) -> (object, chosen indices, result): the SAME tensors are handed out while the choice is unchanged, so the
            # executor's per-conditioning K/V cache and captured graph (keyed on tensor identity) survive between schedule switches


def _memoised(obj, chosen, build):
    hit = _memo.get(id(obj))
    if hit is not None and hit[0] is obj and hit[1] == chosen:
        return hit[2]
    if len(_memo) > 16:
        _memo.clear()
    res = build()
    _memo[id(obj)] = (obj, chosen, res)
    return res


def reconstruct_cond_batch(c, current_step):
    """:294-318.  Ready tensors / dicts (no schedule) pass through; a list of per-image schedules is resolved for `current_step`."""
    if isinstance(c, (torch.Tensor, dict)):
        return _as_cond(c)
    chosen = tuple(_pick(sched, current_step) for sched in c)
    return _memoised(c, chosen, lambda: _reconstruct_cond_batch(c, current_step))


def _reconstruct_cond_batch(c, current_step):
    param = c[0][0].cond
    if isinstance(param, dict):
        res = DictWithShape({k: torch.zeros((len(c),) + v.shape, device=v.device, dtype=v.dtype) for k, v in param.items()})
    else:
        res = torch.zeros((len(c),) + param.shape, device=param.device, dtype=param.dtype)
    for i, cond_schedule in enumerate(c):
        chosen = cond_schedule[_pick(cond_schedule, current_step)].cond
        if isinstance(param, dict):
            for k, v in chosen.items():
                res[k][i] = v
        else:
            res[i] = chosen
    return res


def stack_conds(tensors):
    """:321-334: prompts of different chunk counts are padded by repeating their last token vector."""
    tensors = list(tensors)
    token_count = max(x.shape[0] for x in tensors)
    for i in range(len(tensors)):
        if tensors[i].shape[0] != token_count:
            last_vector = tensors[i][-1:]
            tensors[i] = torch.vstack([tensors[i], last_vector.repeat([token_count - tensors[i].shape[0], 1])])
    return torch.stack(tensors)


def reconstruct_multicond_batch(c, current_step):
    """:337-365 -> (conds_list, stacked conds).  conds_list[i] = [(row of the stacked tensor, weight), ...] for image i: one entry per
    AND-part.  Ready tensors / dicts stand for plain prompts: one (i, 1.0) entry per image."""
    if isinstance(c, (torch.Tensor, dict)):
        cond = _as_cond(c)
        b = cond["crossattn"].shape[0] if isinstance(cond, dict) else cond.shape[0]
        return [[(i, 1.0)] for i in range(b)], cond
    if isinstance(c, list):  # a plain list of per-image schedules (what the reference builds for the NEGATIVE prompt): one part per image
        return [[(i, 1.0)] for i in range(len(c))], reconstruct_cond_batch(c, current_step)
    chosen = tuple(_pick(part.schedules, current_step) for parts in c.batch for part in parts)
    return _memoised(c, chosen, lambda: _reconstruct_multicond_batch(c, current_step))


def _reconstruct_multicond_batch(c, current_step):
    param = c.batch[0][0].schedules[0].cond
    tensors, conds_list = [], []
    for composable_prompts in c.batch:
        conds_for_batch = []
        for part in composable_prompts:
            conds_for_batch.append((len(tensors), part.weight))
            tensors.append(part.schedules[_pick(part.schedules, current_step)].cond)
        conds_list.append(conds_for_batch)
    if isinstance(tensors[0], dict):
        stacked = DictWithShape({k: stack_conds([x[k] for x in tensors]) for k in tensors[0].keys()})
    else:
        stacked = stack_conds(tensors).to(device=param.device, dtype=param.dtype)
    return conds_list, stacked
