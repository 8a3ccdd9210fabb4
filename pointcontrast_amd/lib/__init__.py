"""Host side of the pre-training path: config, samplers, loaders, synthetic pairs, flat parameters / gradient reducer, SGD, trainers."""
