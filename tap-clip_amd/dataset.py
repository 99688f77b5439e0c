"""Few-shot image-folder loaders with the reference's contract, without torchvision.

Mirror of reference dataset.py:8-71: `get_dataloaders(root_dir, class_names, batch_size, num_shots,
preprocess) -> (train_loader | None, val_loader)`; images come from `<root_dir>/<class>/<file>`,
labels are re-indexed to the order of `class_names` (the prompt order), `num_shots` images per class go
to the train split (none for 0: zero-shot, train_loader is None) and up to 100 of the remaining images per
class to validation.  Output batches: float32 `[B, 3, S, S]` (after `preprocess`, e.g.
`CLIPWrapper.get_preprocess()`), int64 `[B]` labels -- what `FullModel.forward(images, labels)` takes
(SURVEY.md section 8f row 3).  `seed` makes the split reproducible (the reference's is unseeded)."""
import os
import random
from collections import defaultdict
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class ImageFolder(Dataset):
    """`<root>/<class name>/<image>`; classes sorted by name like torchvision's ImageFolder."""

    def __init__(self, root: str, transform: Optional[Callable] = None):
        self.root, self.transform = root, transform
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples: List[Tuple[str, int]] = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        from PIL import Image

        path, label = self.samples[idx]
        with open(path, "rb") as fh:
            img = Image.open(fh).convert("RGB")
        return (self.transform(img) if self.transform else img), label


class RelabeledSubset(Dataset):
    """Subset whose raw folder labels are mapped to 0..n-1 in prompt order (reference dataset.py:8-19)."""

    def __init__(self, dataset: Dataset, indices: Sequence[int], raw_to_new_label_map):
        self.dataset, self.indices, self.label_map = dataset, list(indices), raw_to_new_label_map

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        image, raw_label = self.dataset[self.indices[idx]]
        return image, self.label_map[raw_label]


def _raw_u8(img) -> torch.Tensor:
    """decoded RGB image -> uint8 [h, w, 3] (what `engine.preprocess_u8` takes)"""
    import numpy as np

    return torch.from_numpy(np.array(img))


def _collate_raw(batch):
    images, labels = zip(*batch)  # images keep their own sizes: a list, not a stacked tensor
    return list(images), torch.tensor(labels, dtype=torch.int64)


class GpuPreprocessLoader:
    """Wraps a loader of (list of uint8 photos, labels) batches: the eval transform runs on the GPU, one
    `tapclip_preprocess_u8` call per batch, and the loop body sees the same `(images [B,3,S,S] fp32, labels)`
    pairs -- bit-identical values -- as with `preprocess=clip.get_preprocess()` in the workers."""

    def __init__(self, loader: DataLoader, size: int, device="cuda"):
        self.loader, self.size, self.device = loader, size, torch.device(device)
        self.dataset = loader.dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        from . import engine

        for photos, labels in self.loader:
            yield engine.preprocess_u8(photos, size=self.size, device=self.device), labels.to(self.device, non_blocking=True)


def get_dataloaders(root_dir="data/OfficeHomeDataset_10072016/Real_World", class_names=None, batch_size=32,
                    num_shots=5, preprocess=None, *, num_workers: int = 4, seed: Optional[int] = None,
                    gpu_preprocess: Optional[int] = None, device="cuda"):
    """`gpu_preprocess=S` (instead of `preprocess=`): the workers only decode; resize / crop / normalise to S x S
    run on `device` per batch (`GpuPreprocessLoader`)."""
    rng = random.Random(seed) if seed is not None else random
    if gpu_preprocess is not None and preprocess is not None:
        raise ValueError("give either preprocess= (CPU, per sample) or gpu_preprocess=<size>, not both")
    full = ImageFolder(root_dir, transform=_raw_u8 if gpu_preprocess is not None else preprocess)
    raw_to_new = {full.class_to_idx[name]: i for i, name in enumerate(class_names)}  # KeyError for an unknown class, like the reference
    by_label = defaultdict(list)
    for idx, (_, label) in enumerate(full.samples):  # from the sample list: no image is decoded here
        if label in raw_to_new:
            by_label[label].append(idx)

    train_idx: List[int] = []
    if num_shots > 0:
        for label, idxs in by_label.items():
            train_idx.extend(rng.sample(idxs, min(len(idxs), num_shots)))
    else:
        print("[dataset] num_shots=0 -> train set is empty (zero-shot setting)")
    taken = set(train_idx)
    val_idx: List[int] = []
    for label, idxs in by_label.items():
        rest = [i for i in idxs if i not in taken]
        val_idx.extend(rng.sample(rest, min(len(rest), 100)))

    train_set = RelabeledSubset(full, train_idx, raw_to_new)
    val_set = RelabeledSubset(full, val_idx, raw_to_new)
    collate = _collate_raw if gpu_preprocess is not None else None
    train_loader = None if num_shots == 0 else DataLoader(train_set, batch_size=batch_size, shuffle=True, num_workers=num_workers,
                                                          collate_fn=collate)
    val_loader = DataLoader(val_set, batch_size=batch_size, shuffle=False, num_workers=num_workers, collate_fn=collate)
    if gpu_preprocess is not None:
        train_loader = None if train_loader is None else GpuPreprocessLoader(train_loader, gpu_preprocess, device)
        val_loader = GpuPreprocessLoader(val_loader, gpu_preprocess, device)
    print("Raw -> New Label Map:", raw_to_new)
    print("Total Classes (Prompt):", len(class_names))
    return train_loader, val_loader
