"""Host-side indexing of the data plane (integer-exact restatement; no image decoding here).

  * LMDB key scheme and length rule: /root/reference/dataset.py:183,215-216 and utils/make_lmdb.py:63-67
  * per-rank sample order: torch DistributedSampler(shuffle=True) + set_epoch (trainer.py:108-121,230)
  * batch collation with broken-sample / NaN-batch skipping: utils/collate_fn.py:4-17
"""
from __future__ import annotations

import math

import torch


def lmdb_keys(idx: int):
    return f"image-{idx:08d}".encode(), f"meta-{idx:08d}".encode()


def lmdb_length(entries: int) -> int:
    return entries // 2


def next_index_on_decode_failure(idx: int, length: int) -> int:
    """dataset.py:218-222: a corrupt sample is replaced by its successor (wrapping)."""
    return (idx + 1) % length


def distributed_sampler_indices(n, world, rank, epoch, seed=0, drop_last=False):
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    if drop_last and n % world != 0:
        num = math.ceil((n - world) / world)
    else:
        num = math.ceil(n / world)
    total = num * world
    if not drop_last:
        pad = total - len(idx)
        idx += idx[:pad] if pad <= len(idx) else (idx * math.ceil(pad / len(idx)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:world]


def skip_broken_collate_fn(batch):
    batch = [x for x in batch if x is not None]
    if not batch:
        return None
    images = torch.stack([b["image"] for b in batch], dim=0)
    text = [b["text"] for b in batch]
    if torch.isnan(images).any():
        print("[collate_fn] batch has NaN image, skipping batch.")
        return None
    return {"image": images, "text": text}
