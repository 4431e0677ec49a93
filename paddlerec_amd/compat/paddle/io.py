"""paddle.io — Dataset / IterableDataset / DataLoader as tools/utils/utils_single.py:84-108 uses them: an
IterableDataset yields one sample (a list of arrays) at a time, the loader stacks batch_size of them per field
(drop_last) and is CALLED to get a fresh iterator (`train_dataloader()`)."""
import numpy as _np
import torch as _t


class Dataset:
    def __init__(self):
        pass


class IterableDataset(Dataset):
    pass


class DistributedBatchSampler:
    def __init__(self, *a, **k):
        raise NotImplementedError("DistributedBatchSampler: the reference's dygraph trainer never instantiates it")


class DataLoader:
    def __init__(self, dataset, batch_size=1, places=None, drop_last=False, num_workers=0, shuffle=False,
                 return_list=True, collate_fn=None):
        self.dataset, self.batch_size, self.drop_last = dataset, int(batch_size), drop_last
        # places = what paddle.set_device returned (tools/trainer.py:121-126): batches are delivered ON that device, as
        # Paddle's loader does — din/dygraph_model.py:44-54 feeds them to the net without a to_tensor of its own
        place = places[0] if isinstance(places, (list, tuple)) and places else places
        self._device = place if isinstance(place, _t.device) and place.type == "cuda" else None

    def _batches(self):
        buf = []
        it = iter(self.dataset) if hasattr(self.dataset, "__iter__") else (self.dataset[i] for i in range(len(self.dataset)))
        for sample in it:
            buf.append(sample)
            if len(buf) == self.batch_size:
                yield self._collate(buf)
                buf = []
        if buf and not self.drop_last:
            yield self._collate(buf)

    def _collate(self, samples):
        n = len(samples[0])
        out = [_t.from_numpy(_np.stack([_np.asarray(s[i]) for s in samples])) for i in range(n)]
        return out if self._device is None else [t.to(self._device, non_blocking=True) for t in out]

    def __iter__(self):
        return self._batches()

    def __call__(self):
        return self._batches()
