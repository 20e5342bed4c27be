"""The one data helper on the PSO path (reference src/utils/util_data.py:129-135)."""
import torch


def rescale_torch(img, min_val=None, max_val=None):
    """(img - min) / (max - min) over the WHOLE tensor it is given.

    Kept with the reference's quirk: a bound that is falsy (None, or an explicit 0) is
    replaced by the tensor's own min / max.  The swarm loop does not call this: there the
    rescale is per image and fused into the generator's last kernel.
    """
    if not min_val:
        min_val = torch.min(img)
    if not max_val:
        max_val = torch.max(img)
    return (img - min_val) / (max_val - min_val)
