"""Wall-clock meters (pc/lib/timer.py:9-61)."""
import time


class AverageMeter:

  def __init__(self):
    self.reset()

  def reset(self):
    self.val = self.avg = self.sum = self.sq_sum = self.count = 0.0

  def update(self, val, n=1):
    self.val = val
    self.sum += val * n
    self.count += n
    self.avg = self.sum / self.count
    self.sq_sum += val ** 2 * n
    self.var = self.sq_sum / self.count - self.avg ** 2


class Timer:

  def __init__(self):
    self.total_time = self.calls = self.start_time = self.diff = self.avg = 0.0

  def reset(self):
    self.__init__()

  def tic(self):
    self.start_time = time.time()

  def toc(self, average=True):
    self.diff = time.time() - self.start_time
    self.total_time += self.diff
    self.calls += 1
    self.avg = self.total_time / self.calls
    return self.avg if average else self.diff
