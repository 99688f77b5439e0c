"""Accuracy / attribution metrics with the reference's function names and return values
(reference utils/eval_metrics.py:6-96), counting on the device.

The reference walks every sample in Python (`t.item()` per sample: one host sync each,
eval_metrics.py:26-29,60-63); here predictions are compared and binned per batch with two
`bincount`s and the totals are read back once at the end.  The model contract is unchanged:
`model(images)['logits']` is a `[B, n_cls]` float tensor (eval_metrics.py:19-20)."""
from collections import defaultdict

import torch


@torch.no_grad()
def _count(model, dataloader, device):
    """Per-class (correct, total) counts over the loader.

    Data-parallel FullModel (`gather_images=True`): every rank holds the GLOBAL logits (rank-major rows), so the rank's
    labels are gathered the same way and every rank counts the same global totals.  Local batches may differ in length
    (an evaluation loader sharded without padding ends in a short batch on some ranks): embeddings and labels are
    gathered with their row counts exchanged first (`dist.all_gather_rows(ragged=True)`).  Ranks may also run different
    NUMBERS of batches: the loop agrees on the longest shard and feeds empty batches on the ranks that ran out
    (`_synced_batches`; a loader without a reliable length agrees batch by batch).  A sampler that pads its
    shards with DUPLICATE samples (torch's DistributedSampler with drop_last=False) has those duplicates counted: shard
    without padding (e.g. indices[rank::world]) to get the single-process figures."""
    model.eval()
    correct = total = None
    gather = bool(getattr(model, "gather_images", False))
    was_ragged = getattr(model, "ragged_batches", False)
    if gather:
        model.ragged_batches = True
    try:
        correct, total = _count_loop(model, dataloader, device, gather)
    finally:
        if gather:
            model.ragged_batches = was_ragged
    # the counts were just read on the host: the one place where asking the text tower whether its tied-padding claim held
    # costs nothing (FullModel.check_tied_padding raises with the cause; a false claim makes every logit NaN on purpose)
    if hasattr(model, "check_tied_padding"):
        model.check_tied_padding()
    return correct, total


def _synced_batches(model, dataloader, device):
    """The rank's batches, followed by empty ones until the rank with the most batches has run out (data-parallel
    evaluation only: every iteration of the loop body issues collectives that all ranks must join)."""
    import torch.distributed as dist

    from ..dist import is_distributed

    if not is_distributed():
        yield from dataloader
        return
    cpu = dist.get_backend() == "gloo"

    def agree_max(v: int) -> int:
        t = torch.tensor([v], dtype=torch.int64, device="cpu" if cpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    size = getattr(getattr(getattr(model, "clip", None), "cfg", None), "image_size", None)
    shape = None if size is None else (3, size, size)

    def empty():
        if shape is None:
            raise RuntimeError("a rank without a single batch cannot shape its empty batches: the model has no clip.cfg.image_size")
        return torch.zeros((0,) + tuple(shape)), torch.zeros(0, dtype=torch.int64)

    it = iter(dataloader)
    try:  # (every torch DataLoader HAS __len__; over an IterableDataset it raises TypeError or is only an estimate)
        n_mine = len(dataloader)
    except TypeError:
        n_mine = -1
    # ONE collective decides the mode for everybody: a rank without a usable length turns the whole loop to per-batch agreement
    if agree_max(1 if n_mine < 0 else 0) == 0:
        longest = agree_max(n_mine)                # ONE collective for the sized part of the loop
        for _ in range(longest):
            batch = next(it, None)
            if batch is None:
                batch = empty()
            else:
                shape = tuple(batch[0].shape[1:])
            yield batch
        # a length that under-counted (an estimate): keep agreeing until EVERY rank's iterator is exhausted -- batches beyond
        # the agreed length are evaluated, not dropped
    while True:                                    # a loader without a (reliable) length: one tiny collective per batch
        batch = next(it, None)
        if agree_max(0 if batch is None else 1) == 0:
            return
        if batch is None:
            batch = empty()
        else:
            shape = tuple(batch[0].shape[1:])
        yield batch


def _count_loop(model, dataloader, device, gather):
    correct = total = None
    for images, labels in (_synced_batches(model, dataloader, device) if gather else dataloader):
        images, labels = images.to(device), labels.to(device)
        logits = model(images)["logits"]
        n_cls = logits.shape[1]
        if gather:
            from ..dist import all_gather_rows
            labels = all_gather_rows(labels, ragged=True)
            if labels.shape[0] != logits.shape[0]:
                raise RuntimeError(f"gathered {labels.shape[0]} labels for {logits.shape[0]} logit rows: ranks disagree on the batch")
        if labels.numel() == 0:
            continue
        preds = torch.argmax(logits, dim=1)
        hit = (preds == labels).to(torch.int64)
        size = max(n_cls, int(labels.max()) + 1) if correct is None else max(n_cls, correct.numel(), int(labels.max()) + 1)
        c = torch.bincount(labels, weights=hit.double(), minlength=size)
        t = torch.bincount(labels, minlength=size).double()
        if correct is None:
            correct, total = c, t
        else:
            if size > correct.numel():
                pad = size - correct.numel()
                correct = torch.nn.functional.pad(correct, (0, pad))
                total = torch.nn.functional.pad(total, (0, pad))
            correct[: c.numel()] += c
            total[: t.numel()] += t
    if correct is None:
        return torch.zeros(0), torch.zeros(0)
    return correct.cpu(), total.cpu()


def evaluate_accuracy(model, dataloader, device):
    """Overall accuracy in percent; prints the per-class table like the reference."""
    correct, total = _count(model, dataloader, device)
    n = float(total.sum())
    acc = 100.0 * float(correct.sum()) / n if n > 0 else 0.0
    print(f"Overall Accuracy: {acc:.2f}%")
    print("Per-Class Accuracy:")
    for cls in range(total.numel()):
        if total[cls] > 0:
            print(f" - Class {cls:2d}: {100.0 * float(correct[cls]) / float(total[cls]):.2f}% ({int(correct[cls])}/{int(total[cls])})")
    return acc


def evaluate_per_class_accuracy(model, dataloader, device, class_names=None):
    """{class name or index: accuracy %} for the classes that occur."""
    correct, total = _count(model, dataloader, device)
    out = {}
    for cls in range(total.numel()):
        if total[cls] > 0:
            name = class_names[cls] if class_names else str(cls)
            out[name] = 100.0 * float(correct[cls]) / float(total[cls])
    return out


def attribution_entropy(attribution_scores):
    """Mean entropy of the attribution distributions (lower = more concentrated)."""
    p = attribution_scores + 1e-8
    return float((-(p * torch.log(p)).sum(dim=-1)).mean())


def attribution_variance(attribution_scores, labels):
    """Mean over label groups of the per-token variance of their attribution vectors."""
    groups = defaultdict(list)
    for a, l in zip(attribution_scores, labels):
        groups[int(l)].append(a)
    vs = [float(torch.stack(g).var(dim=0).mean()) for g in groups.values()]
    return sum(vs) / len(vs) if vs else 0.0
