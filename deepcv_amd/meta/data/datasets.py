""" Input side of the hot path (mirror of the parts of `src/deepcv/meta/data/datasets.py` `train()` touches).

`dataloader_prefetch_batches` (`datasets.py:76-115`) patches a `DataLoader` so that the next batch is already on its way to the device while the
current one is being trained on.  Here the prefetch is a real double buffer: batches are staged in pinned host memory and copied on a dedicated
HIP copy stream, the compute stream only waits on the copy's event.  The kedro `PytorchDataset` wrapper of the reference
(`datasets.py:30-73`) is project-shell glue and out of scope; `SyntheticImageDataset` is the synthetic-data source named by BASELINE.json. """
import logging
from typing import Any, Iterable, Optional, Sequence, Tuple, Union

import torch
from torch.utils.data import DataLoader, Dataset

__all__ = ['dataloader_prefetch_batches', 'PrefetchedLoader', 'SyntheticImageDataset', 'SyntheticUint8ImageDataset', 'DeviceNormalize']


def _to_device(batch, device, non_blocking=True):
    if isinstance(batch, torch.Tensor):
        return batch.to(device=device, non_blocking=non_blocking)
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to_device(b, device, non_blocking) for b in batch)
    if isinstance(batch, dict):
        return {k: _to_device(v, device, non_blocking) for k, v in batch.items()}
    return batch


def _record_stream(batch, stream):
    if isinstance(batch, torch.Tensor):
        if batch.is_cuda:
            batch.record_stream(stream)
    elif isinstance(batch, (list, tuple)):
        for b in batch:
            _record_stream(b, stream)
    elif isinstance(batch, dict):
        for b in batch.values():
            _record_stream(b, stream)


class DeviceNormalize:
    """ `ToTensor()` + `Normalize(mean, std)` of the reference's preprocessing recipe (`conf/base/parameters.yml:196-201`) applied ON THE DEVICE to a batch
    of raw `[N, H, W, C]` uint8 images (`dcv_images_u8_normalize`): the loader ships a quarter of the bytes over PCIe and no CPU worker converts pixels.
    Returns `[N, C, H, W]` fp32 (NCHW memory, what the first convolution reads in place). """

    def __init__(self, mean: Sequence[float], std: Sequence[float]):
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self._dev = {}

    def __call__(self, images_u8: torch.Tensor) -> torch.Tensor:
        from ... import _native as nat
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
            raise TypeError(f'DeviceNormalize expects a [N, H, W, C] uint8 batch, got {images_u8.dtype} {tuple(images_u8.shape)}')
        nat.require_device(images_u8, 'DeviceNormalize input')
        n, h, w, c = images_u8.shape
        if c != len(self.mean):
            raise ValueError(f'DeviceNormalize: {c} channels but {len(self.mean)} mean values')
        images_u8 = images_u8.contiguous()
        dev = images_u8.device
        if dev not in self._dev:
            self._dev[dev] = (torch.tensor(self.mean, dtype=torch.float32, device=dev), torch.tensor([1.0 / s for s in self.std], dtype=torch.float32, device=dev))
        mean, inv_std = self._dev[dev]
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        nat.check(nat.lib.dcv_images_u8_normalize(images_u8.data_ptr(), out.data_ptr(), n, h, w, c, nat.LAYOUT_NCHW, mean.data_ptr(), inv_std.data_ptr(),
                                                  nat.current_stream(dev)), 'dcv_images_u8_normalize')
        return out


class PrefetchedLoader:
    """ Iterates `dataloader`, keeping one batch in flight to `device` on a side stream.  Exposes `batch_size`, `dataset`, `sampler`, `__len__` of the
    wrapped loader; iterators carry the `_prefetched_batch` attribute the reference's `process_function` probes (`ignite_training.py:235`). """

    def __init__(self, dataloader: DataLoader, device: Union[str, torch.device], device_transform=None):
        self.dataloader = dataloader
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self.device_transform = device_transform     # e.g. `DeviceNormalize`: applied to the first element of every batch on the copy stream

    def __len__(self):
        return len(self.dataloader)

    def __getattr__(self, name):
        return getattr(self.dataloader, name)

    def __iter__(self):
        return _PrefetchIterator(iter(self.dataloader), self.device, self.copy_stream, self.device_transform)


class _PrefetchIterator:
    def __init__(self, it, device, copy_stream, device_transform=None):
        self._it, self._device, self._stream, self._transform = it, device, copy_stream, device_transform
        self._prefetched_batch = None
        self._event = None
        self._fetch()

    def _fetch(self):
        try:
            batch = next(self._it)
        except StopIteration as e:
            self._prefetched_batch = e
            return
        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                self._prefetched_batch = self._apply(_to_device(batch, self._device, non_blocking=True))
                self._event = torch.cuda.Event()
                self._event.record(self._stream)
        else:
            self._prefetched_batch = self._apply(_to_device(batch, self._device))

    def _apply(self, batch):
        if self._transform is None:
            return batch
        if isinstance(batch, (list, tuple)):
            return type(batch)((self._transform(batch[0]), *batch[1:]))
        return self._transform(batch)

    def __iter__(self):
        return self

    def __next__(self):
        if isinstance(self._prefetched_batch, StopIteration):
            raise self._prefetched_batch
        batch = self._prefetched_batch
        if self._event is not None:
            cur = torch.cuda.current_stream(self._device)
            cur.wait_event(self._event)            # compute stream waits for the H2D copy only
            _record_stream(batch, cur)             # keep the allocator from recycling the buffers while compute uses them
        self._fetch()
        return batch


def dataloader_prefetch_batches(dataloader: DataLoader, device: Union[None, str, torch.device] = None, device_transform=None) -> Union[DataLoader, PrefetchedLoader]:
    """ Returns a loader which prefetches the next batch to `device` during compute (reference `datasets.py:76-115`).  Like the reference, nothing is
    patched (the loader is returned as is) when `device` is `None` / CPU; the device is resolved lazily. """
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type == 'cpu':
        logging.warning(f'Warning: DataLoader wont prefetch data batches as given `device` argument is `{device}` when prefetching is aimed at GPU(s).')
        return dataloader
    if not getattr(dataloader, 'pin_memory', False):
        logging.warning(f'Warning: `pin_memory=False`: batches of `{type(dataloader).__name__}` are prefetched through pageable memory (slower H2D copies)')
    return PrefetchedLoader(dataloader, device, device_transform)


class SyntheticImageDataset(Dataset):
    """ `x ~ N(0, 1)` fp32 images of `image_shape` and `y ~ U{0..C-1}` int64 labels drawn once from a seeded host generator (seed 434546 = the
    preprocessing seed of `conf/base/parameters.yml:183`; SURVEY.md §8d "Synthetic inputs"). """

    def __init__(self, size: int, image_shape: Sequence[int] = (3, 32, 32), num_classes: int = 10, seed: int = 434546):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.randn(size, *image_shape, generator=g)
        self.targets = torch.randint(0, num_classes, (size,), generator=g, dtype=torch.int64)
        self.classes = list(range(num_classes))

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, idx) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.images[idx], self.targets[idx]


class SyntheticUint8ImageDataset(Dataset):
    """ Raw-pixel twin of `SyntheticImageDataset`: `[H, W, C]` uint8 images (the storage format of CIFAR / ImageNet decoders) to be normalised on the
    device by `DeviceNormalize`. """

    def __init__(self, size: int, image_shape: Sequence[int] = (3, 32, 32), num_classes: int = 10, seed: int = 434546):
        g = torch.Generator().manual_seed(seed)
        c, h, w = image_shape
        self.images = torch.randint(0, 256, (size, h, w, c), generator=g, dtype=torch.uint8)
        self.targets = torch.randint(0, num_classes, (size,), generator=g, dtype=torch.int64)
        self.classes = list(range(num_classes))

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, idx) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.images[idx], self.targets[idx]
