import torch  # noqa: F401


def denormalize(imgs, dataset):
    """(imgs * std + mean).round().clamp(0, 255) with the ACDC / DSB15 statistics (reference src/utils.py:1-20)."""
    if dataset not in ['acdc', 'dsb15']:
        raise ValueError(f"The name of the dataset should be 'acdc' or 'dsb15'. Got {dataset}.")
    mean, std = (54.089, 48.084) if dataset == 'acdc' else (51.193, 52.671)
    return (imgs.clone() * std + mean).round().clamp(0, 255)
