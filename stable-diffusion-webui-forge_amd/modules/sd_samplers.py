"""`create_sampler` / `add_sampler` -- mirror of modules/sd_samplers.py:34-70."""
from . import sd_samplers_kdiffusion, sd_samplers_lcm, sd_samplers_timesteps
from ..modules_forge import alter_samplers

all_samplers = [*sd_samplers_kdiffusion.samplers_data_k_diffusion, *sd_samplers_timesteps.samplers_data_timesteps,
                *sd_samplers_lcm.samplers_data_lcm, *alter_samplers.samplers_data_alter]
all_samplers_map = {x.name: x for x in all_samplers}


def find_sampler_config(name):
    if name is not None:
        config = all_samplers_map.get(name)
        if config is None:
            for s in all_samplers:
                if name in s.aliases:
                    return s
    else:
        config = all_samplers[0]
    return config


def create_sampler(name, model):
    config = find_sampler_config(name)
    assert config is not None, f"bad sampler name: {name}"
    sampler = config.constructor(model)
    sampler.config = config
    return sampler


def add_sampler(sampler):
    global all_samplers, all_samplers_map
    if sampler.name not in [x.name for x in all_samplers]:
        all_samplers.append(sampler)
        all_samplers_map = {x.name: x for x in all_samplers}
