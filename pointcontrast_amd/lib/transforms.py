"""Feature transforms of the pair loader (pc/lib/transforms.py:10-30): Compose and Jitter
(features += N(mu, sigma) with probability 0.95)."""
import random

import numpy as np


class Compose:

  def __init__(self, transforms):
    self.transforms = list(transforms)

  def __call__(self, coords, feats):
    for t in self.transforms:
      coords, feats = t(coords, feats)
    return coords, feats


class Jitter:

  def __init__(self, mu=0, sigma=0.01, p=0.95):
    self.mu, self.sigma, self.p = mu, sigma, p

  def __call__(self, coords, feats):
    if random.random() < self.p:
      feats = feats + np.random.normal(self.mu, self.sigma, feats.shape)
    return coords, feats
