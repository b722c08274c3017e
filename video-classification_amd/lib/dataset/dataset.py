"""reference lib/dataset/dataset.py:6-12 is an empty function.  The build supplies synthetic
clips of the configured shape (there is no dataset on the box): N(0,1) fp32 clips and
uniform labels, seeded per rank."""
import torch
from torch.utils.data import Dataset


class SyntheticClips(Dataset):
    def __init__(self, length, clip_len=16, crop=112, classes=100, seed=1234):
        self.length, self.shape, self.classes, self.seed = length, (3, clip_len, crop, crop), classes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return torch.randn(self.shape, generator=g), int(torch.randint(0, self.classes, (1,), generator=g))


class SyntheticFrames(Dataset):
    """raw decoded videos: uint8 Ts x Hs x Ws x 3 frames (what a decoder hands over), sampled / cropped / normalised
    on the GPU by utils.ChooseInput (vc_clip_pack_u8) instead of by loader workers."""

    def __init__(self, length, frames=32, height=128, width=171, classes=100, seed=1234):
        self.length, self.shape, self.classes, self.seed = length, (frames, height, width, 3), classes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return (torch.randint(0, 256, self.shape, generator=g, dtype=torch.uint8),
                int(torch.randint(0, self.classes, (1,), generator=g)))


def get_dataset(if_train=True, config=None, length=None):
    from core.config import config as default_config
    cfg = config or default_config
    n = length if length is not None else (256 if if_train else 64)
    return SyntheticClips(n, cfg.MODEL.CLIP_LEN, cfg.MODEL.CROP_SIZE, cfg.MODEL.CLFDIM,
                          seed=1234 if if_train else 4321)


class MultiViewClips(Dataset):
    """validation videos as V views (clips x crops) each: item = (V x 3 x T x H x W, label)."""

    def __init__(self, length, views, clip_len=16, crop=112, classes=100, seed=4321):
        self.length, self.views, self.shape, self.classes, self.seed = length, views, (3, clip_len, crop, crop), classes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return (torch.randn((self.views,) + self.shape, generator=g),
                int(torch.randint(0, self.classes, (1,), generator=g)))
