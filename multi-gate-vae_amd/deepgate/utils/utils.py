"""Small helpers of the reference's utils/utils.py that the training path uses (:14-36)."""
import torch


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        if self.count > 0:
            self.avg = self.sum / self.count


def zero_normalization(x):
    """(x - mean) / unbiased std (utils/utils.py:32-36).  Inside the training step the functional loss
    kernel fuses this; the stand-alone function is for small host-side use."""
    return (x - torch.mean(x)) / torch.std(x)
