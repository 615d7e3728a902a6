"""CPU oracle for the input feeding of the RefineNet hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this file; the shipped path is ``rnh_cine_gather`` (include/refinenet_hip.h) behind
``hipvsr/cine_cache.py``.  numpy restatement of what the reference does per sample and per batch:

* ``RandomHorizontalFlip`` / ``RandomVerticalFlip``   reference ``src/data/transforms.py:321-372``
* ``RandomCropPatch`` (+ ``_get_coordinates``)        reference ``src/data/transforms.py:375-450``
* ``Normalize``                                       reference ``src/data/transforms.py:100-168``
* ``ToTensor``                                        reference ``src/data/transforms.py:74-97``
* ``AcdcVSRRefineNetDataset.__getitem__``             reference ``src/data/datasets/acdc_vsr_refinenet_dataset.py:49-89``
* default collate + ``_get_inputs_targets``           reference ``src/runner/trainers/acdc_vsr_refinenet_trainer.py:64-74``

Parity status: the four transforms are PINNED against the reference's own classes (``tests/golden/g6_input.pt``,
written by ``tests/golden/make_golden.py`` which imports ``src/data/transforms.py`` in the build container; its
``import SimpleITK`` line needs a placeholder module there, none of the classes exercised touches it).  The window
arithmetic of ``__getitem__`` is restated from the cited lines and NOT pinned by execution: the dataset class reads its
cines through nibabel, which this image does not have - "parity unpinned" for that part.

The draws of the augmentation are explicit arguments here; ``draw_augment`` consumes a ``random.Random`` in the
reference's order (flip, flip, crop row, crop column - the order of the ``augments`` list in exp1_x4.yaml:17-23).
"""
import numpy as np


def draw_augment(rng, lr_shape, size, hflip_prob=0.5, vflip_prob=0.5):
    """(hflip, vflip, h0, w0) drawn like the reference: ``random.random() < prob`` for each flip (transforms.py:344,371),
    then ``random.randint(0, h - ht)``, ``random.randint(0, w - wt)`` on the LR image (transforms.py:443-444)."""
    hflip = rng.random() < hflip_prob
    vflip = rng.random() < vflip_prob
    h, w = lr_shape[:2]
    ht, wt = size
    if h - ht < 0 or w - wt < 0:
        raise ValueError(f'The image ({lr_shape}) is smaller than the cropped size ({size}). Please use a smaller cropped size.')
    return hflip, vflip, rng.randint(0, h - ht), rng.randint(0, w - wt)


def augment(imgs, hflip, vflip, h0, w0, size, ratio):
    """imgs: LR frames then HR frames, each (H, W, C).  transforms.py:345 (flip axis 1), :372 (flip axis 0), :409-416."""
    if hflip:
        imgs = [np.flip(im, 1) for im in imgs]
    if vflip:
        imgs = [np.flip(im, 0) for im in imgs]
    half = len(imgs) // 2
    lr, hr = imgs[:half], imgs[half:]
    if not all(j // i == ratio for a, b in zip(lr, hr) for i, j in zip(a.shape[:-1], b.shape[:-1])):
        raise ValueError(f'The ratio between the HR images and the LR images should be {ratio}.')
    hn, wn = h0 + size[0], w0 + size[1]
    return [im[h0:hn, w0:wn] for im in lr] + [im[h0 * ratio:hn * ratio, w0 * ratio:wn * ratio] for im in hr]


def normalize(img, means, stds):
    """transforms.py:154-168: per channel (x - mean) / (std + 1e-10), in the array's own dtype."""
    img = img.copy()
    for c, mean, std in zip(range(img.shape[-1]), means, stds):
        img[..., c] = (img[..., c] - mean) / (std + 1e-10)
    return img


def get_item(lr_cine, hr_cine, code, t, num_frames, num_updated_frames, draws=None, size=None, ratio=None, means=None,
             stds=None):
    """One sample.  lr_cine / hr_cine: (H, W, C, T) arrays as nib.load(...).get_data() returns them (dataset :54-55),
    code: (T,) phase code of the patient (:66-71).  ``t`` = target frame for a training sample, None for the whole cycle
    (valid / test).  Returns lists of (C, h, w) float32 arrays and the (F, 1) code."""
    Tc = lr_cine.shape[-1]
    imgs = [lr_cine[..., k] for k in range(Tc)] + [hr_cine[..., k] for k in range(hr_cine.shape[-1])]     # :56-57
    if t is not None and draws is not None:                                                                 # :59-60
        imgs = augment(imgs, *draws, size, ratio)
    if means is not None:
        imgs = [normalize(im, means, stds) for im in imgs]                                                  # :61, Normalize
    imgs = [np.ascontiguousarray(np.asarray(im, dtype=np.float32).transpose(2, 0, 1)) for im in imgs]       # ToTensor, :62
    lr, hr = imgs[:len(imgs) // 2], imgs[len(imgs) // 2:]
    code = np.asarray(code).astype(np.float32)                                                              # :71 (not normalised)
    lr, hr = lr + lr + lr, hr + hr + hr                                                                     # :74
    code = np.tile(code, 3)[:, None]                                                                        # :75
    T3 = len(lr) // 3
    U = num_updated_frames
    if t is not None:                                                                                       # :78-84
        tt = t + T3
        start, end = tt - num_frames + 1, tt + 1
        return lr[start - U:end + U], hr[start:end], code[start - U:end + U]
    return lr[T3 - U:2 * T3 + U], hr[:T3], code[T3 - U:2 * T3 + U]                                          # :85-88


def collate(samples):
    """Default collate of N samples + the trainer's view of it: inputs list[F] of (N, C, h, w), targets list[T],
    pos_codes (N, F, 1)."""
    F, T = len(samples[0][0]), len(samples[0][1])
    inputs = [np.stack([s[0][k] for s in samples]) for k in range(F)]
    targets = [np.stack([s[1][i] for s in samples]) for i in range(T)]
    pos = np.stack([s[2] for s in samples])
    return inputs, targets, pos
