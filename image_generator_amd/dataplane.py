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


# ------------------------------------------------------------------------------------------------
# LMDB reader + sample decoding (/root/reference/dataset.py:169-229)
# ------------------------------------------------------------------------------------------------
class LMDBEnv:
    """Read-only LMDB environment on the library's page walker (`mi355_lmdb_*`; liblmdb / py-lmdb are not needed):
    what dataset.py:173-183 uses of `lmdb.open(readonly=True, lock=False)` - `entries` and `get(key)`."""

    def __init__(self, path):
        import ctypes
        from ._lib import lib
        self._lib, self._ct = lib, ctypes
        self._env = lib.mi355_lmdb_open(str(path).encode())
        if not self._env:
            raise OSError(f"LMDBEnv: {lib.mi355_last_error().decode()}")

    @property
    def entries(self) -> int:
        return int(self._lib.mi355_lmdb_entries(self._env))

    def get(self, key: bytes):
        """bytes of the value, or None when the key is absent (as `txn.get` does)."""
        ct = self._ct
        data, size = ct.c_void_p(), ct.c_longlong()
        rc = self._lib.mi355_lmdb_get(self._env, key, len(key), ct.byref(data), ct.byref(size))
        if rc == 1:
            return None
        if rc != 0:
            raise OSError(f"LMDBEnv.get: {self._lib.mi355_last_error().decode()}")
        return ct.string_at(data.value, size.value)

    def close(self):
        if getattr(self, "_env", None):
            self._lib.mi355_lmdb_close(self._env)
            self._env = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_metadata(blob: bytes) -> dict:
    """The metadata record of make_lmdb.py:22-30 is `pickle.dumps({'raw_text': str})`.  It is read with an unpickler that
    resolves NO globals: plain containers and strings load, anything that would import or call code raises."""
    import io
    import pickle

    class _NoGlobals(pickle.Unpickler):
        def find_class(self, module, name):
            raise pickle.UnpicklingError(f"metadata record refers to {module}.{name}: only plain dict / str records are read")

    meta = _NoGlobals(io.BytesIO(blob)).load()
    if not isinstance(meta, dict) or not isinstance(meta.get("raw_text"), str):
        raise ValueError("metadata record is not {'raw_text': str}")
    return meta


def _adjust_hue(img, hue_factor):
    """torchvision.transforms.functional.adjust_hue for PIL images [EXT torchvision 0.21]: shift the H channel of the
    HSV image by uint8(hue_factor * 255) with wrap-around."""
    import numpy as np
    from PIL import Image
    mode = img.mode
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = (np_h.astype(np.int32) + (int(hue_factor * 255) & 0xFF)).astype(np.uint8)     # uint8 arithmetic wraps
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert(mode)


def train_transform(image, size=512):
    """dataset.py:186-199 on a PIL RGB image -> float tensor [3, size, size] in [-1, 1]:
    Resize((size, size)) bilinear -> ColorJitter(.02, .02, .02, .01) with probability 0.2 -> horizontal flip with
    probability 0.5 -> ToTensor -> Normalize(0.5, 0.5).  The random draws are made from torch's global generator in the
    order the torchvision transforms make them [EXT torchvision 0.21: RandomApply `p < rand(1)` skips; ColorJitter draws
    randperm(4) then the four factors; RandomHorizontalFlip `rand(1) < p` flips], so a seeded run picks the same
    augmentations; the pixel arithmetic is PIL's own resize / ImageEnhance, which is what torchvision calls for PIL input."""
    import numpy as np
    from PIL import Image, ImageEnhance
    image = image.resize((size, size), Image.BILINEAR)
    if not (0.2 < float(torch.rand(1))):
        order = torch.randperm(4).tolist()
        b = float(torch.empty(1).uniform_(0.98, 1.02))
        c = float(torch.empty(1).uniform_(0.98, 1.02))
        s = float(torch.empty(1).uniform_(0.98, 1.02))
        h = float(torch.empty(1).uniform_(-0.01, 0.01))
        for fn in order:
            if fn == 0:
                image = ImageEnhance.Brightness(image).enhance(b)
            elif fn == 1:
                image = ImageEnhance.Contrast(image).enhance(c)
            elif fn == 2:
                image = ImageEnhance.Color(image).enhance(s)
            else:
                image = _adjust_hue(image, h)
    if float(torch.rand(1)) < 0.5:
        image = image.transpose(Image.FLIP_LEFT_RIGHT)
    x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    return x.sub_(0.5).div_(0.5)


def eval_transform(image, size=224):
    """dataset.py:201-209: Resize((224, 224)) -> ToTensor -> Normalize(0.5, 0.5)."""
    import numpy as np
    from PIL import Image
    image = image.resize((size, size), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    return x.sub_(0.5).div_(0.5)


class LMDBImageDataset(torch.utils.data.Dataset):
    """Drop-in for /root/reference/dataset.py:169-229: same constructor, `__len__` (= entries // 2), `__getitem__`
    returning {"image": float tensor in [-1, 1], "text": {"raw_text": str}}, the same key scheme and the same
    fall-through to the next sample when an image does not decode."""

    def __init__(self, lmdb_dir, is_train=True):
        self.lmdb_dir = lmdb_dir
        self.is_train = is_train
        self.env = LMDBEnv(lmdb_dir)
        self.length = lmdb_length(self.env.entries)

    def __len__(self):
        return self.length

    def __getstate__(self):         # DataLoader workers re-open the map instead of inheriting a C pointer
        return {"lmdb_dir": self.lmdb_dir, "is_train": self.is_train, "length": self.length}

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.env = LMDBEnv(self.lmdb_dir)

    def __getitem__(self, idx):
        import io
        from PIL import Image
        k_img, k_meta = lmdb_keys(idx)
        img_bytes, meta_bytes = self.env.get(k_img), self.env.get(k_meta)
        try:
            image = Image.open(io.BytesIO(img_bytes)).convert("RGB")
        except Exception as e:
            print(e)
            return self.__getitem__(next_index_on_decode_failure(idx, len(self)))
        raw_text = load_metadata(meta_bytes)["raw_text"]
        image = train_transform(image) if self.is_train else eval_transform(image)
        return {"image": image, "text": {"raw_text": raw_text}}
